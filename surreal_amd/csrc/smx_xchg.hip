// Data-parallel exchange between the learner ranks of one node over IPC-mapped peer buffers (xGMI loads),
// the collectives of surreal_amd.learner: what must cross ranks for N sharded learners to equal
// the single reference learner -- the per-epoch gradient / loss-partial sum (surreal/learner/ppo.py:541-562 run on
// shards), the advantage moments (ppo.py:413-416) and the end-of-learn statistics.
//
// Why not RCCL for these: the exchange is 2.1 MB, ten times per learn, on the critical path of a 1.2 ms step.  A
// ring all-reduce is 2 (W - 1) dependent hops of a library kernel launched eagerly between graph segments; here it is
// ONE kernel of the learner's own graph (capturable: plain launches, the sequence number lives in device memory):
//
//   two-shot all-reduce (W ranks, vector split into W chunks, grid of NB workgroups; workgroup w owns slice w of
//   every chunk, so a workgroup only ever depends on the SAME-index workgroup of its peers):
//     0  copy own input -> own staging S;                     release flag[0][w] = seq
//     1  wait flag[0][w] of every peer; chunk `rank` = sum over ranks 0..W-1 IN RANK ORDER of the peers' S (pulled
//        over xGMI, W loads in flight per lane -- one per link); -> own R[seq & 1] and own `out`;
//                                                                release flag[1][w] = seq
//     2  wait flag[1][w] of every peer; pull their reduced chunks from their R[seq & 1] -> own `out`
//   Every rank gets the bits rank c computed for chunk c: replicas stay bit-identical by construction.
//   one-shot all-gather (small payloads): own part -> own R[seq & 1]; release flag[0]; wait; pull.
//
// Memory protocol.  A flag is written only by its owner, into its OWN buffer, with a system-scope release after the
// data it guards; peers POLL IT REMOTELY with system-scope loads and take a system-scope acquire before they touch the
// data.  Nobody ever polls local memory for a remote write.  Flags carry the exchange's sequence number (monotone,
// wrap-safe compare): nothing is ever reset.  Buffer reuse needs no extra handshake: S is overwritten by the next
// exchange's step 0, and a rank leaves step 2 only after every peer released flag[1] -- i.e. finished reading S; R is
// double-buffered by sequence parity, and a rank cannot pass the first wait of exchange seq + 1 before every peer has
// completed exchange seq (stream order), so R[seq & 1] is free again at seq + 2.
// Every spin is bounded by the wall clock (s_memrealtime, 100 MHz): a timeout raises an error word in the rank's own
// buffer and in a caller-supplied device word (read back with the learner's statistics); later waits return at once.
//
// The buffers are allocated HERE (the one exception to the ABI's "nothing allocates": torch's caching allocator cannot
// hand out an IPC-exportable, fine-grained allocation), uncached / fine-grained so that peer reads do not depend on L2
// write-back, exported with hipIpcGetMemHandle and opened by the peers once per learner workspace.
#include "smx_common.h"
#include <string.h>

namespace {

constexpr int XW = SMX_XCHG_MAX_RANKS;
constexpr int XB = 64;               // workgroups per exchange at most (flag columns)
constexpr int XTH = 512;
constexpr int FLAG_STRIDE = 16;      // one 64-byte line per flag
constexpr int CTRL_BYTES = 256;      // seq | ticket | err | timeout ticks (lo, hi)
constexpr int FLAG_BYTES = 2 * XB * FLAG_STRIDE * 4;
constexpr int HDR_BYTES = 16384;
static_assert(CTRL_BYTES + FLAG_BYTES <= HDR_BYTES, "header");

struct Ctrl {
    unsigned seq;        // sequence number of the last completed exchange
    int ticket;
    unsigned err;
    unsigned pad;
    long long timeout;   // wall-clock ticks (100 MHz) a wait may take
};

// Layout.  A vector of n floats is cut into W chunks of `chunk` = ceil(n / W) rounded UP to XAL floats; element e of the
// vector sits at S[e], so the staging must hold W * chunk floats -- up to W * XAL - 1 more than n (zero padding that step
// 0 really writes).  Sizing S by the capacity alone let that padding run into the head of R[0] (12 floats at 4 ranks, 28
// at 8): harmless unless a late step-0 workgroup's padding landed after a sibling's step-1 result -- a wrong sum once
// in many runs under contention.  Chunks, workgroup slices, S and both halves of R are whole 256-byte lines, so no line
// is ever shared by two writers or two phases.
constexpr long XAL = 64;
__host__ __device__ inline long round_up(long v, long a) { return (v + a - 1) / a * a; }
__host__ __device__ inline long chunk_of(long n, int world) { return round_up((n + world - 1) / world, XAL); }
__host__ __device__ inline long chunk_cap(long capacity, int world) { return chunk_of(capacity, world); }
__host__ __device__ inline long staging_floats(long capacity, int world) { return world * chunk_cap(capacity, world); }
__host__ __device__ inline long total_bytes(long capacity, int world) {
    return HDR_BYTES + 4L * (staging_floats(capacity, world) + 2 * chunk_cap(capacity, world));
}

struct View {
    Ctrl* ctrl;
    unsigned* flags;     // [2][XB][FLAG_STRIDE]
    float* S;
    float* R;            // [2][chunk_cap]
};
__device__ __forceinline__ View view(void* base, long capacity, int world) {
    char* p = (char*)base;
    View v;
    v.ctrl = (Ctrl*)p;
    v.flags = (unsigned*)(p + CTRL_BYTES);
    v.S = (float*)(p + HDR_BYTES);
    v.R = v.S + staging_floats(capacity, world);
    return v;
}
__device__ __forceinline__ unsigned* flag_of(const View& v, int phase, int w) {
    return v.flags + ((size_t)phase * XB + w) * FLAG_STRIDE;
}

__device__ __forceinline__ long long wall() { return (long long)wall_clock64(); }

// one lane per peer polls that peer's flag (remote, system scope) until it reaches `seq`; the workgroup then takes a
// system-scope acquire.  Returns false (and raises the error words) on a timeout; once an error is up nobody waits.
__device__ __forceinline__ bool wait_peers(const smx_xchg_t& X, const View& own, int phase, int w, unsigned seq,
                                           int* err_out) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    const int p = threadIdx.x;
    if (p < X.world && p != X.rank) {
        const View pv = view(X.peer[p], X.capacity, X.world);
        const unsigned* f = flag_of(pv, phase, w);
        const long long t0 = wall(), lim = own.ctrl->timeout;
        bool good = true;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (__hip_atomic_load(&own.ctrl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                wall() - t0 > lim) {
                good = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) {
            const unsigned code = 0x100u | ((unsigned)phase << 4) | (unsigned)p;     // timeout | phase | peer
            atomicOr(&own.ctrl->err, code);
            if (err_out) atomicOr(err_out, (int)code);
            ok = 0;
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");    // "" = system scope
    return ok != 0;
}

__device__ __forceinline__ void release_flag(const View& own, int phase, int w, unsigned seq) {
    // EVERY wavefront takes a SYSTEM-scope release behind its own stores (s_waitcnt vmcnt(0) + write-back: a wavefront's
    // fence only covers that wavefront's stores, and the workgroup barrier does not wait on vmcnt -- with lane 0 alone
    // fencing, the other seven wavefronts' stores to S / R were not formally ordered in front of the flag), then the
    // barrier, then lane 0 publishes the flag with a system-scope release store
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // "" = system scope
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(flag_of(own, phase, w), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the exchange's last workgroup publishes the new sequence number (the next launch reads it at its start)
__device__ __forceinline__ void finish(const View& own, unsigned seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // every wavefront's stores to `out` are out before the ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(&own.ctrl->ticket, 1);
        if (t == (int)gridDim.x - 1) {
            own.ctrl->ticket = 0;
            own.ctrl->seq = seq;
            __threadfence();
        }
    }
}

__device__ __forceinline__ float4 ld4f(const float* p) { return *(const float4*)p; }

__global__ __launch_bounds__(XTH) void xchg_allreduce_kernel(smx_xchg_t X, const float* in,
                                                            float* out, long n, int* err_out) {
    const View own = view(X.peer[X.rank], X.capacity, X.world);
    const unsigned seq = own.ctrl->seq + 1u;
    const int W = X.world, w = blockIdx.x, NB = gridDim.x;
    const long chunk = chunk_of(n, W);
    const long sub = round_up((chunk + NB - 1) / NB, XAL);
    const long s0 = (long)w * sub, s1 = s0 + sub < chunk ? s0 + sub : chunk;       // this workgroup's slice of a chunk
    // ---- 0: own input -> S ------------------------------------------------------------------------
    for (int c = 0; c < W; ++c) {
        const long base = (long)c * chunk;
        for (long i = s0 + 4L * threadIdx.x; i < s1; i += 4L * XTH) {
            const long e = base + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e + 3 < n) v = ld4f(in + e);
            else {
                if (e < n) v.x = in[e];
                if (e + 1 < n) v.y = in[e + 1];
                if (e + 2 < n) v.z = in[e + 2];
            }
            *(float4*)(own.S + e) = v;
        }
    }
    release_flag(own, 0, w, seq);
    // ---- 1: reduce chunk `rank` over the ranks, in rank order ---------------------------------------
    // a wait that timed out leaves `out` alone from here on (what it would sum is not what the peers meant to send);
    // the flags are still published so that a peer which is merely late is not held up as well, and the optimiser
    // launch that follows skips its step when the error word is up (clip_adam_kernel)
    const bool ok0 = wait_peers(X, own, 0, w, seq, err_out);
    const float* Sp[XW];
#pragma unroll
    for (int p = 0; p < XW; ++p) Sp[p] = view(X.peer[p < W ? p : 0], X.capacity, X.world).S;
    float* Rmine = own.R + (size_t)(seq & 1u) * chunk_cap(X.capacity, W);
    {
        const long base = (long)X.rank * chunk;
        for (long i = s0 + 4L * threadIdx.x; ok0 && i < s1; i += 4L * XTH) {
            float4 v[XW];
#pragma unroll
            for (int p = 0; p < XW; ++p)
                if (p < W) v[p] = ld4f(Sp[p] + base + i);            // W loads in flight, one per link
            float4 a = v[0];
#pragma unroll
            for (int p = 1; p < XW; ++p)
                if (p < W) { a.x += v[p].x; a.y += v[p].y; a.z += v[p].z; a.w += v[p].w; }
            *(float4*)(Rmine + i) = a;
            const long e = base + i;
            if (e + 3 < n) *(float4*)(out + e) = a;
            else {
                if (e < n) out[e] = a.x;
                if (e + 1 < n) out[e + 1] = a.y;
                if (e + 2 < n) out[e + 2] = a.z;
            }
        }
    }
    release_flag(own, 1, w, seq);
    // ---- 2: pull the other ranks' reduced chunks ------------------------------------------------------
    const bool ok1 = wait_peers(X, own, 1, w, seq, err_out) && ok0;
    const float* Rp[XW];
#pragma unroll
    for (int p = 0; p < XW; ++p)
        Rp[p] = view(X.peer[p < W ? p : 0], X.capacity, X.world).R + (size_t)(seq & 1u) * chunk_cap(X.capacity, W);
    for (long i = s0 + 4L * threadIdx.x; ok1 && i < s1; i += 4L * XTH) {
        float4 v[XW];
#pragma unroll
        for (int p = 0; p < XW; ++p)
            if (p < W && p != X.rank) v[p] = ld4f(Rp[p] + i);
#pragma unroll
        for (int p = 0; p < XW; ++p)
            if (p < W && p != X.rank) {
                const long e = (long)p * chunk + i;
                if (e + 3 < n) *(float4*)(out + e) = v[p];
                else {
                    if (e < n) out[e] = v[p].x;
                    if (e + 1 < n) out[e + 1] = v[p].y;
                    if (e + 2 < n) out[e + 2] = v[p].z;
                }
            }
    }
    finish(own, seq);
}

__global__ __launch_bounds__(XTH) void xchg_allgather_kernel(smx_xchg_t X, const float* __restrict__ in, long n_per,
                                                            float* __restrict__ out, int* err_out) {
    const View own = view(X.peer[X.rank], X.capacity, X.world);
    const unsigned seq = own.ctrl->seq + 1u;
    const int W = X.world, w = blockIdx.x, NB = gridDim.x;
    long sub = (n_per + NB - 1) / NB;
    const long s0 = (long)w * sub, s1 = s0 + sub < n_per ? s0 + sub : n_per;
    const size_t roff = (size_t)(seq & 1u) * chunk_cap(X.capacity, W);
    for (long i = s0 + threadIdx.x; i < s1; i += XTH) own.R[roff + i] = in[i];
    release_flag(own, 0, w, seq);
    const bool ok = wait_peers(X, own, 0, w, seq, err_out);
    for (int p = 0; ok && p < W; ++p) {
        const float* src = view(X.peer[p], X.capacity, X.world).R + roff;
        for (long i = s0 + threadIdx.x; i < s1; i += XTH) out[(long)p * n_per + i] = src[i];
    }
    // (flag[1] keeps pace: a later all-reduce's step 2 compares against ITS sequence number only)
    finish(own, seq);
}

__global__ void xchg_init_kernel(void* base, long long timeout_ticks) {
    Ctrl* c = (Ctrl*)base;
    c->seq = 0; c->ticket = 0; c->err = 0; c->pad = 0; c->timeout = timeout_ticks;
}

__global__ void xchg_status_kernel(const void* base, unsigned* out) {
    const Ctrl* c = (const Ctrl*)base;
    out[0] = c->seq; out[1] = c->err;
}

int check(const smx_xchg_t* x) {
    SMX_REQUIRE(x, SMX_E_NULL);
    SMX_REQUIRE(x->world >= 2 && x->world <= XW && x->rank >= 0 && x->rank < x->world && x->capacity > 0, SMX_E_SHAPE);
    for (int p = 0; p < x->world; ++p) SMX_REQUIRE(x->peer[p], SMX_E_NULL);
    return SMX_OK;
}

int blocks_for(long floats) {
    long b = (floats + 2047) / 2048;            // one 16-byte word per lane and pass
    return (int)(b < 1 ? 1 : (b > XB ? XB : b));
}

}  // namespace

extern "C" int64_t smx_xchg_bytes(int64_t capacity_floats, int32_t world) {
    if (capacity_floats <= 0 || world < 2 || world > XW) return 0;
    return total_bytes(capacity_floats, world);
}

extern "C" int smx_xchg_alloc(int64_t bytes, double timeout_s, void** ptr, int32_t* kind, smx_stream_t stream) {
    SMX_REQUIRE(ptr, SMX_E_NULL);
    SMX_REQUIRE(bytes >= HDR_BYTES, SMX_E_SHAPE);
    void* p = nullptr;
    int k = 0;                                        // 0: uncached, 1: fine-grained, 2: plain hipMalloc
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        k = 1;
        if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            k = 2;
            const hipError_t e = hipMalloc(&p, (size_t)bytes);
            if (e != hipSuccess) return (int)e;
        }
    }
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, smx_s(stream));
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    const long long ticks = (long long)((timeout_s > 0 ? timeout_s : 2.0) * 1e8);    // s_memrealtime: 100 MHz
    hipLaunchKernelGGL(xchg_init_kernel, dim3(1), dim3(1), 0, smx_s(stream), p, ticks);
    e = hipStreamSynchronize(smx_s(stream));
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    *ptr = p;
    if (kind) *kind = k;
    return SMX_OK;
}

extern "C" int smx_xchg_free(void* ptr) {
    SMX_REQUIRE(ptr, SMX_E_NULL);
    return (int)hipFree(ptr);
}

extern "C" int smx_xchg_export(void* ptr, void* handle64) {
    SMX_REQUIRE(ptr && handle64, SMX_E_NULL);
    static_assert(sizeof(hipIpcMemHandle_t) == SMX_XCHG_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle64, &h, sizeof(h));
    return SMX_OK;
}

extern "C" int smx_xchg_open(const void* handle64, void** ptr) {
    SMX_REQUIRE(handle64 && ptr, SMX_E_NULL);
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

extern "C" int smx_xchg_close(void* ptr) {
    SMX_REQUIRE(ptr, SMX_E_NULL);
    return (int)hipIpcCloseMemHandle(ptr);
}

extern "C" int smx_xchg_allreduce_f32(const smx_xchg_t* x, const float* in, float* out, int64_t n, int32_t* err,
                                      smx_stream_t stream) {
    const int rc = check(x);
    if (rc) return rc;
    SMX_REQUIRE(in && out, SMX_E_NULL);
    SMX_REQUIRE(n > 0 && n <= x->capacity, SMX_E_SHAPE);
    SMX_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, SMX_E_ALIGN);
    const long chunk = chunk_of(n, x->world);
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(blocks_for(chunk)), dim3(XTH), 0, smx_s(stream), *x, in, out, (long)n,
                       (int*)err);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_xchg_allgather_f32(const smx_xchg_t* x, const float* in, int64_t n_per_rank, float* out, int32_t* err,
                                      smx_stream_t stream) {
    const int rc = check(x);
    if (rc) return rc;
    SMX_REQUIRE(in && out, SMX_E_NULL);
    SMX_REQUIRE(n_per_rank > 0 && n_per_rank <= chunk_cap(x->capacity, x->world), SMX_E_SHAPE);
    hipLaunchKernelGGL(xchg_allgather_kernel, dim3(blocks_for(n_per_rank / 4 + 1)), dim3(XTH), 0, smx_s(stream), *x, in,
                       (long)n_per_rank, out, (int*)err);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

extern "C" int smx_xchg_status(const smx_xchg_t* x, uint32_t* seq_err_dev, smx_stream_t stream) {
    const int rc = check(x);
    if (rc) return rc;
    SMX_REQUIRE(seq_err_dev, SMX_E_NULL);
    hipLaunchKernelGGL(xchg_status_kernel, dim3(1), dim3(1), 0, smx_s(stream), (const void*)x->peer[x->rank], seq_err_dev);
    SMX_LAUNCH_CHECK();
    return SMX_OK;
}

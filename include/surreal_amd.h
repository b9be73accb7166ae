/*
 * surreal_amd.h -- C ABI of libsurreal_amd.so: the MI355X (gfx950) hot path of
 * SurrealAI/surreal's Agent -> Replay -> Learner data path.
 *
 * The reference is 100 % Python on PyTorch ATen ops: it has NO native layer and
 * NO FFI for this path.  The entry points below are
 * therefore exactly the set a maintainer's ctypes binding would call from the
 * reference's own plugin classes; each one cites the reference interface
 * (file:line, relative to the reference tree) whose arithmetic it replaces.
 * INTEGRATION.md shows the reference-side ctypes stubs.
 *
 * Conventions
 *   - extern "C", plain device pointers + sizes, no torch / C++ types.
 *   - every function returns 0 on success, a negative SMX_E_* argument error, or a
 *     positive hipError_t from the launch; nothing throws.
 *   - nothing allocates, nothing synchronises: the caller (PyTorch-ROCm on the
 *     host side) owns every buffer, including the workspaces whose sizes the
 *     *_ws_bytes() helpers return.  All launches go to `stream` (a hipStream_t
 *     passed as void*), so every entry point is legal inside hipGraph capture.
 *   - all tensors are dense row-major fp32 unless stated.
 *   - mutable training scalars (learning rates, beta, clip epsilon, Adam step
 *     counters, the KL early-exit flag) live in DEVICE memory (smx_ppo_ctrl) so
 *     that a captured graph can be replayed while they change.
 */
#ifndef SURREAL_AMD_H
#define SURREAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* smx_stream_t; /* hipStream_t */

enum {
    SMX_OK = 0,
    SMX_E_NULL = -1,      /* required pointer is NULL */
    SMX_E_SHAPE = -2,     /* non-positive / inconsistent dimension */
    SMX_E_UNSUPPORTED = -3, /* shape outside what the kernel is built for */
    SMX_E_WORKSPACE = -4, /* workspace too small */
    SMX_E_ALIGN = -5      /* pointer not 16-byte aligned where required */
};

enum { SMX_ACT_NONE = 0, SMX_ACT_RELU = 1, SMX_ACT_TANH = 2 };
enum { SMX_PPO_CLIP = 0, SMX_PPO_ADAPT = 1 };

int smx_abi_version(void);
const char* smx_error_string(int code);

/* ---------------------------------------------------------------------------
 * Three-layer MLP: Linear(D,H1)-ReLU-Linear(H1,H2)-ReLU-Linear(H2,OUT)[-Tanh]
 * = PPO_ActorNetwork / PPO_CriticNetwork (surreal/model/model_builders/builders.py:86-175).
 * W* are [out_features, in_features] row-major (torch.nn.Linear layout).
 * ------------------------------------------------------------------------- */
typedef struct {
    const float* W1; const float* b1; /* [H1,D]  [H1]  */
    const float* W2; const float* b2; /* [H2,H1] [H2]  */
    const float* W3; const float* b3; /* [OUT,H2] [OUT] */
    int32_t D, H1, H2, OUT;
} smx_mlp3_t;

/* --- z-filter (surreal/model/z_filter.py:44-79) ---------------------------- */
/* mean = sum/count ; std = clamp(sqrt(sumsq/count - mean^2), min=eps)  (z_filter.py:74-76) */
int smx_zfilter_stats_f32(const float* running_sum, const float* running_sumsq,
                          const float* count, int32_t D, float eps,
                          float* mean_out, float* std_out, smx_stream_t stream);
/* out[r,:] = clamp((x[r*ldx + :] - mean)/std, -5, 5), r < rows  (z_filter.py:77).
 * ldx = row stride of x in floats (>= D; e.g. N*D to read step 0 of every sub-trajectory,
 * ppo.py:537); out is dense [rows, D]. */
int smx_zfilter_forward_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                            const float* mean, const float* std, float* out,
                            smx_stream_t stream);
/* the two calls above in one launch, straight from the running sums: what an acting agent
 * needs for its one observation per step (PPOAgent.act -> forward_actor -> ZFilter.forward,
 * ppo_agent.py:106-154, z_filter.py:59-79).  Bit-identical to stats + forward. */
int smx_zfilter_forward_sums_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                                 const float* running_sum, const float* running_sumsq,
                                 const float* count, float eps, float* out, smx_stream_t stream);
/* sum += sum_rows x ; sumsq += sum_rows x*x ; count += count_rows  (z_filter.py:55-57).
 * count_rows lets a data-parallel rank add its local column sums while the caller
 * all-reduces them (pass 0 and add the global count once). */
int smx_zfilter_update_f32(const float* x, int64_t ldx, int64_t rows, int32_t D,
                           float* running_sum, float* running_sumsq, float* count,
                           float count_rows, smx_stream_t stream);
/* The same for MANY rows (the z-update of a policy on a stem runs over B * E ~ 10^5 observation rows): the rows are cut
 * into chunks whose column sums go through ws (smx_zfilter_update_ws_floats(rows, D) floats; 0 = few rows) and are
 * added in chunk order by a second launch.  With ws == NULL or too small this IS smx_zfilter_update_f32. */
int64_t smx_zfilter_update_ws_floats(int64_t rows, int32_t D);
int smx_zfilter_update_ws_f32(const float* x, int64_t ldx, int64_t rows, int32_t D, float* running_sum,
                              float* running_sumsq, float* count, float count_rows, float* ws, int64_t ws_floats,
                              smx_stream_t stream);

/* Process-wide switch of smx_mlp3_forward_fused_f32's z-filter arithmetic.  0 (default): (x - m) * (1 / s), one rounding
 * more than surreal/model/z_filter.py:77 (<= 1 ulp of the filtered input); 1: the reference's division, on the
 * generic (slower) staging path.  tests/test_gpu_kernels.py::test_fused_exact_zfilter_switch pins both. */
int smx_mlp3_fused_exact_zfilter(int32_t on);

/* --- fused critic/actor forward over every step of every sub-trajectory ------
 * Replaces PPOModel.forward_critic / forward_actor on the concatenated
 * (obs, obs_next) tensor in PPOLearner._gae_and_return (surreal/learner/ppo.py:376-386,
 * surreal/model/ppo_net.py:253-315): z-filter prologue + three GEMMs on FP32 MFMA
 * with activations kept in registers.  Logical row r = g*(T0+T1)+t reads
 * x_main[g, t, :] for t < T0 and x_tail[g, t-T0, :] otherwise, so the reference's
 * torch.cat([obs, obs_next], dim=1) copy is never materialised.
 *   x_main [G, T0, D], x_tail [G, T1, D] (T1 may be 0, x_tail NULL)
 *   zmean/zstd [D] or NULL (no z-filter)
 *   packed: weights repacked by smx_mlp3_pack_f32 (K-chunked, zero padded)
 *   out [G*(T0+T1), OUT]; out_act SMX_ACT_NONE (critic) or SMX_ACT_TANH (actor mean)
 * Supported: H1 <= 320, H2 <= 224, OUT <= 32, any D >= 1. */
size_t smx_mlp3_packed_bytes(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_mlp3_pack_f32(const smx_mlp3_t* net, float* packed, size_t packed_bytes,
                      smx_stream_t stream);
/* the same launch also refreshing the z-filter statistics the pass that follows reads (exactly
 * smx_zfilter_stats_f32: mean_out / std_out [D] from the running sums) */
int smx_mlp3_pack_zstats_f32(const smx_mlp3_t* net, float* packed, size_t packed_bytes,
                             const float* running_sum, const float* running_sumsq, const float* count,
                             int32_t D, float eps, float* mean_out, float* std_out, smx_stream_t stream);
int smx_mlp3_forward_fused_f32(const float* packed, int32_t D, int32_t H1, int32_t H2,
                               int32_t OUT, const float* x_main, const float* x_tail,
                               int64_t G, int32_t T0, int32_t T1, const float* zmean,
                               const float* zstd, float* out, int32_t out_act,
                               smx_stream_t stream);

/* --- one dense layer on FP32 MFMA (small-batch epochs) ------------------------
 * C[M,N] = act(A[M,K] . B[N,K]^T + bias[N]); every epoch-loop GEMM of
 * _clip_update/_adapt_update/_value_update (ppo.py:227-353) is an instance.
 * a_kcontig: A(m,k) = A[m*lda+k] (1) or A[k*lda+m] (0); same for B.
 * relu_mask (optional, [M,ldc]): C *= (relu_mask > 0)  -- ReLU backward. */
int smx_linear_f32(const float* A, int32_t lda, int32_t a_kcontig, const float* B,
                   int32_t ldb, int32_t b_kcontig, const float* bias, float* C,
                   int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t act,
                   const float* relu_mask, const int32_t* stop_flag, smx_stream_t stream);

/* Up to 9 INDEPENDENT dense problems in ONE launch -- the layers (or weight gradients) of one dependency level of an
 * update, e.g. DDPG's target-actor, critic and actor first layers (surreal/learner/ddpg.py:244-352 issues them as
 * separate ATen calls).  kind 0: C [M,N] = act(A . B^T + bias) (* (relu_mask > 0)), arguments as smx_linear_f32;
 * kind 1: C = dW [M,N] = A^T . B with A = dZ [K rows, >= M] (stride lda), B = X [K rows, >= N] (stride ldb),
 * dbias [M] = column sums of dZ (nullable), arguments as smx_linear_wgrad_f32. */
typedef struct smx_linear_job {
    int32_t kind, act;
    const float* A; const float* B; const float* bias; const float* relu_mask;
    float* C; float* dbias;
    int32_t lda, ldb, ldc, a_kcontig, b_kcontig, M, N, K;
    const int32_t* stop_flag;
} smx_linear_job_t;
int smx_linear_multi_f32(const smx_linear_job_t* jobs, int32_t njobs, smx_stream_t stream);

/* Weight gradient of one dense layer: dW[M,N] = dZ^T . X, db[M] = column sums of dZ (db may be
 * NULL); dZ [rows, .] with row stride ldz, X [rows, .] with row stride ldx (what
 * loss.backward() produces for nn.Linear, ddpg.py:308,331). */
int smx_linear_wgrad_f32(const float* dZ, int32_t ldz, const float* X, int32_t ldx, float* dW,
                         int32_t ldw, float* db, int32_t M, int32_t N, int32_t rows,
                         smx_stream_t stream);

/* The same with the rows cut into chunks (split-K) when rows >> 1000 -- the weight gradients of
 * the LSTM / CNN stems run over B*T or B*E*pixels rows.  ws: caller's workspace of at least
 * smx_linear_wgrad_ws_floats(M, N, rows) floats (0 = the plain entry point is used); partial
 * tiles are added in a fixed order (deterministic).  ws == NULL falls back to the plain call. */
int64_t smx_linear_wgrad_ws_floats(int32_t M, int32_t N, int32_t rows);
int smx_linear_wgrad_splitk_f32(const float* dZ, int32_t ldz, const float* X, int32_t ldx,
                                float* dW, int32_t ldw, float* db, int32_t M, int32_t N,
                                int32_t rows, float* ws, int64_t ws_floats, smx_stream_t stream);
/* TWO weight gradients from the same dZ [rows, M] (the LSTM's dW_ih = dgates^T . x and dW_hh = dgates^T . h_prev,
 * ppo_net.py:143-152 under loss.backward()): dW1 [M, N1] = dZ^T . X1, dW2 [M, N2] = dZ^T . X2 (dense, row stride N),
 * db1 / db2 [M] column sums (nullable).  One split-K launch for both when both run on the 32 x 32-tile kernel
 * (ws >= smx_linear_wgrad_ws_floats(M, N1, rows) + ..(M, N2, rows) floats), otherwise exactly two
 * smx_linear_wgrad_splitk_f32 calls; results are those of the two calls, bit for bit. */
int smx_linear_wgrad_splitk_pair_f32(const float* dZ, int32_t ldz, int32_t M, int32_t rows, const float* X1,
                                     int32_t ldx1, float* dW1, float* db1, int32_t N1, const float* X2,
                                     int32_t ldx2, float* dW2, float* db2, int32_t N2, float* ws,
                                     int64_t ws_floats, smx_stream_t stream);

/* One MLP forward or backward "job" for the multi-network entry points below: PPO's actor and
 * critic are independent networks updated in lock-step epochs (ppo.py:541-562), so their
 * layer-l GEMMs share one launch.  Forward uses net/x/rows/h1/h2/out/out_act; backward uses
 * net/x/rows/h1/h2/dz3 (in) and dz2/dz1/grads/sumsq_partials (out).  stop_flag (device int,
 * may be NULL): non-zero turns this job's part of every launch into a no-op. */
typedef struct {
    const smx_mlp3_t* net;
    const float* x;
    int64_t rows;
    float* h1;
    float* h2;
    float* out;
    int32_t out_act;
    int32_t out_ld;      /* row stride of `out` in floats; 0 = OUT (dense) */
    const float* dz3;
    float* dz2;
    float* dz1;
    float* grads;
    float* sumsq_partials;
    const int32_t* stop_flag;
    /* optional transposed copies, all [features, rows] with row stride ldT floats (>= rows; 0 means
     * rows -- pad it, e.g. rows + 16: a power-of-two stride puts all 32 rows of a fragment load
     * on the same cache set / memory channel): the forward writes
     * h1T / h2T, the backward writes dz2T / dz1T; when xT, dz3T and all four are given the
     * weight-gradient GEMMs read K-contiguous operands (16-byte loads) instead of strided ones */
    float* h1T;
    float* h2T;
    const float* xT;
    const float* dz3T;
    float* dz2T;
    float* dz1T;
    int64_t ldT;
} smx_mlp3_job_t;
/* forward: 1 <= njobs <= 4 (one launch per layer for all jobs; jobs may differ in rows and
 * shapes -- e.g. actor, critic, reference actor and the critic over the obs_next rows at the
 * start of a learn); backward: 1 <= njobs <= 3 (three launches: dz2, dz1, all weight gradients) */
int smx_mlp3_forward_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs, smx_stream_t stream);
int smx_mlp3_backward_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs, smx_stream_t stream);

/* MLP forward keeping the hidden activations (needed by the backward):
 * h1 [rows,H1], h2 [rows,H2], out [rows,OUT] = act(layer 3). */
int smx_mlp3_forward_f32(const smx_mlp3_t* net, const float* x, int64_t rows, float* h1,
                         float* h2, float* out, int32_t out_act,
                         const int32_t* stop_flag, smx_stream_t stream);
/* The same over MANY rows (the MLPs on top of an LSTM / CNN stem: rows = B * E ~ 10^5 per epoch; the reference calls
 * the same nn.Sequential, surreal/model/ppo_net.py:284-315): ONE launch of the fused 16-row kernel -- x read once,
 * activations from the accumulators to h1 / h2 once and on to the next layer in registers -- behind a repack of the
 * weights into `packed` (smx_mlp3_packed_bytes(D, H1, H2, OUT) bytes, 16-byte aligned; contents are scratch).
 * out [rows, OUT] with row stride out_ld floats (0 = OUT).  Returns SMX_E_UNSUPPORTED outside the kernel's fast path
 * (smx_mlp3_forward_rows_supported: D, H1, H2 multiples of 4, 64 < H1 <= 320, 64 < H2 <= 224, OUT <= 32; x, h1, h2
 * 16-byte aligned) -- the caller then uses smx_mlp3_forward_f32.  Results equal the layered path's within fp32
 * rounding (another summation order), not bit for bit. */
int32_t smx_mlp3_forward_rows_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_mlp3_forward_rows_f32(const smx_mlp3_t* net, const float* x, int64_t rows, float* h1, float* h2,
                              float* out, int32_t out_act, int32_t out_ld, float* packed, size_t packed_bytes,
                              const int32_t* stop_flag, smx_stream_t stream);

/* MLP backward from dz3 = dLoss/d(pre-activation of layer 3) [rows,OUT]:
 *   grads  flat [H1*D + H1 + H2*H1 + H2 + OUT*H2 + OUT] in (W1,b1,W2,b2,W3,b3) order
 *   dz2 [rows,H2], dz1 [rows,H1] scratch; sumsq_partials [smx_mlp3_backward_partials()]
 *   receives per-tile sums of squares of the gradients (for clip_grad_norm_). */
int32_t smx_mlp3_backward_partials(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_mlp3_backward_f32(const smx_mlp3_t* net, const float* x, const float* h1,
                          const float* h2, const float* dz3, int64_t rows, float* dz2,
                          float* dz1, float* grads, float* sumsq_partials,
                          const int32_t* stop_flag, smx_stream_t stream);
/* The same for MANY rows (the MLP on top of an LSTM / CNN stem runs over B x T rows, loss.backward() of
 * surreal/learner/ppo.py:227-353 with surreal/model/ppo_net.py:143-152 in front): the rows of every weight gradient are cut
 * into chunks (one workgroup per (tile, chunk), partial tiles in ws), one segmented reduce forms the six gradients in a
 * fixed order.  ws: smx_mlp3_backward_ws_floats() floats (0: few rows, no split); with ws == NULL or too small this IS
 * smx_mlp3_backward_f32 without sum-of-squares partials. */
int64_t smx_mlp3_backward_ws_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT, int64_t rows);
int smx_mlp3_backward_splitk_f32(const smx_mlp3_t* net, const float* x, const float* h1, const float* h2,
                                 const float* dz3, int64_t rows, float* dz2, float* dz1, float* grads, float* ws,
                                 int64_t ws_floats, const int32_t* stop_flag, smx_stream_t stream);
/* ... with the three data-gradient products as ONE fused launch in front of the split-K weight gradients
 * (csrc/smx_mlp3_bwd16.hip: a wavefront owns 16 rows from dz3 to dx, dz2 / dz1 go from the accumulators to memory once
 * and on in registers; the weight gradients of the hidden layers run with the whole dW in one workgroup's registers,
 * csrc/smx_wgrad.hip).  dx [rows, D] (may be NULL): the gradient with respect to the MLP's input -- what the layered
 * path leaves to a separate smx_linear_f32(dz1, W1) call.  packedT: scratch of smx_mlp3_dgrad_rows_ws_floats() floats,
 * 16-byte aligned.  Returns SMX_E_UNSUPPORTED -- nothing launched -- outside the fused kernel's shapes
 * (smx_mlp3_dgrad_rows_supported: D <= 128, D / H1 / H2 multiples of 4, 64 < H1 <= 320, 64 < H2 <= 224, OUT <= 32),
 * for unaligned operands, or when rows are too few for a split-K workspace: the caller then uses the call above.
 * Same sums in another order: equal within fp32 rounding, not bit for bit. */
int32_t smx_mlp3_dgrad_rows_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int64_t smx_mlp3_dgrad_rows_ws_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_mlp3_backward_rows_f32(const smx_mlp3_t* net, const float* x, const float* h1, const float* h2,
                               const float* dz3, int64_t rows, float* dz2, float* dz1, float* dx, float* grads,
                               float* ws, int64_t ws_floats, float* packedT, int64_t packedT_floats,
                               const int32_t* stop_flag, smx_stream_t stream);

/* --- windowed GAE / n-step returns (surreal/learner/ppo.py:387-418) ----------
 * values [B,N+1] RAW critic outputs -- or, when values_tail != NULL, values [B,N] for the N
 * steps and values_tail [B] for obs_next (lets the caller evaluate the two row sets with
 * different kernels); the done-mask values[:,1:] *= 1-dones (ppo.py:387) is applied inside.  H = N -> non-RNN branch (E = 1);
 * H = rnn.horizon -> RNN branch, E = N-H+1 sliding windows.
 * gamma_pow/lam_pow [H] = torch.pow(gamma|lam, arange) as the reference builds
 * them (ppo.py:372-374); gamma_H = gamma**H.
 * adv [B,E] (un-normalised), ret [B,E]. */
int smx_windowed_gae_returns_f32(const float* values, const float* values_tail,
                                 const float* rewards,
                                 const float* dones, const float* gamma_pow,
                                 const float* lam_pow, float gamma, float gamma_H,
                                 int32_t B, int32_t N, int32_t H, float* adv, float* ret,
                                 smx_stream_t stream);
/* smx_windowed_gae_returns_f32 + smx_moments_f32 + smx_adv_normalize_f32 in ONE launch (single rank):
 * the last workgroup to finish forms the batch moments of adv (adv_moments[3]) and normalises adv in
 * place with max(std_unbiased, min_std).  ticket: one int32 in device memory, zero before the first
 * call (the kernel leaves it zero). */
int smx_windowed_gae_norm_f32(const float* values, const float* values_tail, const float* rewards,
                              const float* dones, const float* gamma_pow, const float* lam_pow,
                              float gamma, float gamma_H, int32_t B, int32_t N, int32_t H, float* adv,
                              float* ret, float* adv_moments, float min_std, int32_t* ticket,
                              smx_stream_t stream);
/* rewards * reward_scale and RewardFilter.forward + RewardFilter.update (surreal/learner/ppo.py:452-455,
 * surreal/model/reward_filter.py:33-57) in ONE launch, legal under graph capture:
 *   x = rewards[i] * scale;  out[i] = use_filter ? clamp((x - mean) / std, -5, 5) : x   with
 *   mean = state[1] / state[0], std = max(sqrt(state[2] / state[0] - mean^2), eps) taken BEFORE the update;
 *   update_state: state[0] += n, state[1] += sum(x), state[2] = sum(x*x)  (assigned, reward_filter.py:42).
 * state = {count, running_sum, running_sumsq} (3 floats in device memory); sums (nullable): {n, sum(x),
 * sum(x*x)} of this call (what several ranks exchange before they update the state themselves);
 * partials: smx_reward_filter_partials() doubles of scratch; ticket: one int32, zero before the first call
 * (left zero).  out may alias rewards. */
int smx_reward_filter_f32(const float* rewards, int64_t n, float scale, int32_t use_filter, float* state,
                          float eps, int32_t update_state, float* out, float* sums, double* partials,
                          int32_t* ticket, smx_stream_t stream);
int32_t smx_reward_filter_partials(void);
/* What PPOLearner._optimize does after its epoch loops (ppo.py:565-584) in ONE launch (single rank):
 *   smx_value_loss_finalize_f32 (v_partials [n_epochs, nblk, 8] -> v_stats rows; n_epochs may be 0),
 *   smx_moments_f32 over the return targets (ret_moments[3]), smx_zfilter_update_f32 on x [rows, D]
 *   (x == NULL: no z-filter), then -- by the last workgroup to finish, so it reads the updated sums --
 *   smx_ppo_final_stats_f32 (out4).  ticket as above. */
typedef struct smx_learn_epilogue {
    const float* x; int64_t ldx; int64_t rows; int32_t D; int32_t A;
    float* running_sum; float* running_sumsq; float* count; float count_rows; int32_t n_epochs;
    const float* ret; int64_t n_ret; float* ret_moments;
    const float* v_partials; int32_t nblk; int32_t stats_stride; float* v_stats;
    const float* log_var; float* out4; int32_t* ticket;
} smx_learn_epilogue_t;
int smx_ppo_learn_epilogue_f32(const smx_learn_epilogue_t* args, smx_stream_t stream);
/* moments[3] = {n, mean, M2 = sum (x-mean)^2} over n values (two-pass, one workgroup) */
int smx_moments_f32(const float* x, int64_t n, float* moments, smx_stream_t stream);
/* Chan-merge k per-rank moment triples [k,3] into out[3] (multi-GPU advantage norm) */
int smx_moments_merge_f32(const float* parts, int32_t k, float* out, smx_stream_t stream);
/* x = (x - mean) / max(std_unbiased, min_std)   (ppo.py:402-405, 413-416) */
int smx_adv_normalize_f32(float* x, int64_t n, const float* moments, float min_std,
                          smx_stream_t stream);

/* --- PPO device-resident control block ---------------------------------------
 * Mutable scalars read by the loss / optimiser kernels (see file header). */
typedef struct {
    float lr_actor, lr_critic;  /* current learning rates (ppo.py:576) */
    float beta, eta;            /* adapt mode: KL penalty, cutoff coeff (ppo.py:108-112) */
    float clip_eps;             /* clip mode (ppo.py:115) */
    float kl_target;            /* ppo.py:95 */
    float actor_max_norm, critic_max_norm; /* <=0: no clipping (ppo.py:154-157) */
    float actor_weight_decay, critic_weight_decay;
    int32_t adam_step_actor, adam_step_critic; /* torch.optim.Adam 'step' state */
    int32_t stop_flag;     /* set when KL(ref||curr) > 4*kl_target (ppo.py:556-557) */
    int32_t epochs_done;   /* policy epochs actually applied this learn() */
    int32_t reserved[2];   /* [0]: error word of a peer exchange (smx_xchg_*: pass its address as `err`); [1]: raised by
                              smx_epoch_fwdbwd_f32 when its in-launch wait timed out.  Either makes smx_clip_adam* skip
                              the step. */
} smx_ppo_ctrl_t;

/* per-epoch statistics slots (floats) written by the kernels; see ppo.py:219-224,278-284 */
enum {
    SMX_PS_SURR = 0,     /* _surr_loss */
    SMX_PS_LOSS = 1,     /* _clip_surr_loss | _kl_loss_adapt */
    SMX_PS_ENTROPY = 2,  /* _entropy */
    SMX_PS_KL = 3,       /* mean KL(ref||learn) at this forward */
    SMX_PS_GRADNORM = 4, /* grad_norm_actor */
    SMX_PS_LB = 5,       /* mean behave likelihood */
    SMX_PS_ISW = 6,      /* mean L_learn/(L_behave+1e-4) */
    SMX_PS_REFBEH = 7,   /* mean KL(ref||behave) */
    SMX_PS_STRIDE = 8
};
enum { SMX_VS_LOSS = 0, SMX_VS_EXPVAR = 1, SMX_VS_GRADNORM = 2, SMX_VS_STRIDE = 4 };

/* --- DiagGauss surrogate losses, forward + backward to dz3 --------------------
 * Replaces DiagGauss.loglikelihood/likelihood/kl/entropy (ppo_net.py:29-72) and
 * _clip_loss / _adapt_loss (ppo.py:194-285) including their autograd backward down to
 * the pre-tanh output of the actor's last layer and to log_var.
 *   mean [rows,A] = tanh output of the actor; log_var [A]; actions [rows, lda_act..]
 *   behave [rows,2A] = [mean|std] (row stride ld_beh), ref [rows,2A] (stride ld_ref)
 *   adv [rows] (row stride 1)
 *   row_partials [nblk, SMX_LOSS_PARTIALS(A)], nblk = smx_ppo_loss_blocks(rows)
 *   g_surr, g_kl [rows,A]: d(sum_r surr_r)/dz3 and d(sum_r KL_r)/dz3, combined by
 *   smx_ppo_loss_finalize_f32 once the batch means are known.
 * n_total = global number of rows over all ranks (the mean's denominator). */
int32_t smx_ppo_loss_blocks(int64_t rows);
int32_t smx_ppo_loss_partial_stride(int32_t A);
int smx_ppo_policy_loss_f32(int32_t mode, const float* mean, const float* log_var,
                            const float* actions, int32_t ld_act, const float* behave,
                            int32_t ld_beh, const float* ref, int32_t ld_ref,
                            const float* adv, int64_t rows, int32_t A,
                            const smx_ppo_ctrl_t* ctrl, float* g_surr, float* g_kl,
                            float* row_partials, smx_stream_t stream);
/* Reduce the partials (optionally already all-reduced across ranks: pass nblk = 1 and a
 * summed partial row), write stats[SMX_PS_*], the combined dz3 = (g_surr + c*g_kl)/n_total
 * and the log_var gradient.  When `check_stop` is non-zero it evaluates the KL early-exit
 * test of the PREVIOUS update (mean KL(ref||curr) > 4*kl_target, ppo.py:553-557) and raises
 * ctrl->stop_flag.  When `will_update` is non-zero and the flag stays clear it advances
 * ctrl->adam_step_actor and ctrl->epochs_done for the optimiser step that follows.
 * dlogvar_sumsq (optional) receives sum(dlogvar^2), one more clip_grad_norm_ partial.
 * dz3_t (optional) receives the transposed copy dz3_t[a * ld_t + r] (see smx_mlp3_job_t).
 * No-op when ctrl->stop_flag is already set on entry. */
/* Folds nblk partial rows (row stride = stride floats) to nout rows in front of smx_ppo_loss_finalize_f32: out[j] = the
 * sum of rows [j R, (j + 1) R), R = ceil(nblk / nout), in a fixed order.  For the policies on a stem, whose B * E ~ 10^5
 * rows leave thousands of partial rows that every workgroup of the finalize would walk.  No-op when ctrl (may be NULL)
 * has its stop flag set. */
int smx_ppo_partials_fold_f32(const float* row_partials, int32_t nblk, int32_t stride, float* out, int32_t nout,
                              const smx_ppo_ctrl_t* ctrl, smx_stream_t stream);
int smx_ppo_loss_finalize_f32(int32_t mode, const float* row_partials, int32_t nblk,
                              const float* g_surr, const float* g_kl, const float* log_var,
                              int64_t rows, int64_t n_total, int32_t A, smx_ppo_ctrl_t* ctrl,
                              int32_t check_stop, int32_t will_update, float* dz3,
                              float* dz3_t, int64_t ld_t, float* dlogvar, float* dlogvar_sumsq,
                              float* stats, smx_stream_t stream);

/* --- value loss (ppo.py:311-332): loss = mean((V-ret)^2), explained variance ---
 * dz3[r] = 2 (V_r - ret_r) / n_total ; partials [nblk, 8] = per-block
 * {n, mean_d, M2_d, mean_ret, M2_ret, sum d^2, 0, 0} with d = ret - V (mergeable moments).
 * Advances ctrl->adam_step_critic when will_update != 0. */
int32_t smx_value_loss_blocks(int64_t rows);
int smx_value_loss_f32(const float* values, const float* returns, int64_t rows,
                       int64_t n_total, float* dz3, float* partials, smx_ppo_ctrl_t* ctrl,
                       int32_t will_update, smx_stream_t stream);
/* The three entry points above in one call, for a single-GPU lock-step epoch (n_total == rows;
 * with several ranks the loss partials must be all-reduced between the loss and its finalize, so
 * the separate entry points are used): the policy loss and the value loss share ONE launch
 * (disjoint workgroup ranges), the policy finalize follows.  Results are bit-identical to the
 * three separate calls.  Field meanings as there; values == NULL skips the value loss. */
typedef struct smx_ppo_losses {
    int32_t mode, A;
    const float* mean;
    const float* log_var;
    const float* actions;
    const float* behave;
    const float* ref;
    const float* adv;
    int32_t ld_act, ld_beh, ld_ref, check_stop;
    int64_t rows;
    float* g_surr;
    float* g_kl;
    float* row_partials;
    float* dz3;
    float* dz3_t;
    int64_t ld_t;
    float* dlogvar;
    float* dlogvar_sumsq;
    float* stats;
    const float* values;
    const float* returns;
    float* v_dz3;
    float* v_partials;
    int32_t will_update, v_will_update;
} smx_ppo_losses_t;
int smx_ppo_epoch_losses_f32(const smx_ppo_losses_t* args, smx_ppo_ctrl_t* ctrl,
                             smx_stream_t stream);
/* Data-parallel lock-step epoch (several ranks): ONE all-reduce per epoch instead
 * of one for the loss sums and one for the gradients.  The gradient of either loss is linear in
 * dz3 = (g_surr + c_kl * g_kl) / n_total, and only c_kl needs the GLOBAL mean KL (ppo.py:272-276),
 * so the backward pass runs on the two right-hand sides separately and the combination happens
 * after the all-reduce:
 *   smx_ppo_epoch_losses_dp_f32   the launch of smx_ppo_epoch_losses_f32 WITHOUT its finalize:
 *                                 args->g_surr / g_kl receive the tiles already divided by
 *                                 n_total (+ transposed copies [A, ld_t] when g_surr_t != NULL),
 *                                 args->row_partials the block sums, the value loss uses n_total;
 *                                 args->dz3 / dz3_t / dlogvar / dlogvar_sumsq / stats are ignored
 *   (backward on both right-hand sides; all-reduce of [G_surr | G_critic | G_kl | row_partials])
 *   smx_ppo_epoch_combine_f32     grads_a[i] += c_kl * grads_kl[i] for i < n_mlp (adapt; clip:
 *                                 grads_kl may be NULL), grads_a[n_mlp + a] = log_var's gradient,
 *                                 stats / KL early exit / step counters exactly as
 *                                 smx_ppo_loss_finalize_f32, and the sum-of-squares partials of
 *                                 both groups for smx_clip_adam_step_pair_f32:
 *                                 sumsq_a [smx_sumsq_blocks(n_a)], sumsq_c [smx_sumsq_blocks(n_c)]
 *                                 (grads_c == NULL: actor only). */
int smx_ppo_epoch_losses_dp_f32(const smx_ppo_losses_t* args, int64_t n_total, float* g_surr_t,
                                float* g_kl_t, smx_ppo_ctrl_t* ctrl, smx_stream_t stream);
typedef struct smx_ppo_combine {
    int32_t mode, A;
    const float* row_partials; /* [nblk, 8 + 2A], summed over ranks */
    int32_t nblk;
    int32_t check_stop;
    int64_t n_total;
    const float* log_var;
    float* stats;
    float* grads_a;
    const float* grads_kl;
    int64_t n_mlp, n_a;
    float* sumsq_a;
    const float* grads_c;
    int64_t n_c;
    float* sumsq_c;
    int32_t will_update, reserved;
} smx_ppo_combine_t;
int smx_ppo_epoch_combine_f32(const smx_ppo_combine_t* args, smx_ppo_ctrl_t* ctrl,
                              smx_stream_t stream);
/* the weight-gradient launch of smx_mlp3_backward_multi_f32 alone (all three layers of up to 3 jobs
 * in ONE launch): dW_l = dz_l^T . input_l, db_l = column sums, per-tile sums of squares.  Needs the
 * transposed operands (xT, h1T, h2T, dz3T, dz2T, dz1T, ldT) the fused epoch kernels write. */
int smx_mlp3_wgrad_multi_f32(const smx_mlp3_job_t* jobs, int32_t njobs, smx_stream_t stream);

/* --- fused row-block epoch kernels -------------------------------------------------------------
 * One policy / value epoch of PPOLearner._optimize (surreal/learner/ppo.py:541-562; losses
 * :194-353; forward_actor / forward_critic surreal/model/ppo_net.py:253-315) on a few thousand
 * rows as FOUR dependent launches instead of nine: a workgroup owns 16 rows of one network and runs
 *   smx_epoch_forward_f32   layer 1 -> 2 -> 3 (FP32 MFMA 16x16x4, activations through LDS) and the
 *                           job's loss on those rows: SMX_EPOCH_LOSS_POLICY = DiagGauss likelihoods /
 *                           KL / surrogate -> loss->g_surr, g_kl [rows, A] and the block partial
 *                           sums loss->row_partials [smx_epoch_blocks(rows), 8 + 2A];
 *                           SMX_EPOCH_LOSS_VALUE = loss->v_dz3 [rows] = 2 (V - ret) / n_total and
 *                           loss->v_partials [smx_epoch_blocks(rows), 8] (mergeable moments, the
 *                           layout smx_value_loss_finalize_f32 reads).  h1T / h2T ([H, ldT]) receive
 *                           the hidden activations transposed; out ([rows, OUT], row stride out_ld)
 *                           the network output (optional).  A job with a raised stop_flag is skipped.
 *   smx_epoch_backward_f32  policy job: batch means from the partial rows -> KL coefficient
 *                           (ppo.py:272-276), loss->stats / dlogvar / dlogvar_sumsq, the KL early
 *                           exit and step counters exactly as smx_ppo_loss_finalize_f32, then
 *                           dz3 = (g_surr + c_kl g_kl) / n_total; value job: dz3 = job.dz3 [rows].
 *                           dz2 = (dz3 . W3) * relu'(h2), dz1 = (dz2 . W2) * relu'(h1) written
 *                           transposed (dz3T [OUT, ldT] policy only, dz2T, dz1T).  With
 *                           loss->will_update == 0 (the final, forward-only pass) only the
 *                           statistics / early-exit part runs (one job, one workgroup).
 * followed by smx_mlp3_wgrad_multi_f32 and smx_clip_adam_step_pair_f32.
 * Requirements (smx_epoch_supported): H1, H2 multiples of 4, OUT <= 32, x dense [rows, D]; W2, W3,
 * b1, b2 (and x when D % 4 == 0) 16-byte aligned.  loss->mean / dz3 / dz3_t / values are not used (the tiles stay on chip). */
/* backward only, data-parallel epochs (several ranks): SMX_EPOCH_RHS_SURR / _KL run the actor's data
 * gradients on ONE right-hand side each, dz3 = g_surr / n_total or g_kl / n_total, with no batch means
 * and no statistics (the loss gradient is linear in dz3; smx_ppo_epoch_combine_f32 forms G_surr + c_kl
 * G_kl after the all-reduce) */
enum { SMX_EPOCH_LOSS_NONE = 0, SMX_EPOCH_LOSS_POLICY = 1, SMX_EPOCH_LOSS_VALUE = 2, SMX_EPOCH_RHS_SURR = 3,
       SMX_EPOCH_RHS_KL = 4 };
typedef struct smx_epoch_job {
    const smx_mlp3_t* net;
    const float* x;
    int64_t rows;
    float* h1T;
    float* h2T;
    int64_t ldT;
    float* out;
    int32_t out_ld;   /* 0 = OUT */
    int32_t out_act;
    int32_t loss;     /* SMX_EPOCH_LOSS_* */
    int32_t reserved;
    const int32_t* stop_flag;
    const float* dz3; /* backward, value job: [rows] */
    float* dz3T;
    float* dz2T;
    float* dz1T;
    const float* packed; /* forward: the net's weights as smx_epoch_pack_f32 lays them out */
} smx_epoch_job_t;
/* The forward kernel reads the weights in MFMA fragment order: [tile of 16 features][32-wide K chunk]
 * [half][lane][4], zero padded to whole tiles and an even chunk count, so that every load instruction
 * reads one contiguous KB (from the row-major matrices a fragment load touches 16 cache lines and the
 * loop runs at half the MFMA rate).  smx_epoch_pack_f32 writes that copy for up to 4 networks in one
 * launch; the epoch loop keeps it current by repacking after every optimiser step. */
typedef struct smx_epoch_pack {
    const smx_mlp3_t* net;
    float* packed; /* smx_epoch_packed_floats(D, H1, H2, OUT) floats, 16-byte aligned */
} smx_epoch_pack_t;
int64_t smx_epoch_packed_floats(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_epoch_pack_f32(const smx_epoch_pack_t* items, int32_t n, smx_stream_t stream);
/* What PPOLearner._optimize prepares once per learn for its epoch loops (ppo.py:527-539), in ONE launch:
 *   xn  [rows, D] = ZFilter.forward(obs0) with the model's statistics (zmean / zstd from
 *        smx_zfilter_stats_f32; NULL: plain copy), xnT [D, ldT] its transposed copy (optional);
 *   xr  [rows, D] = the same through the REFERENCE policy's filter, straight from its running sums
 *        (ref_filter != 0; else plain copy)  -- ref_target_model.forward_actor's input (ppo.py:539);
 *   xnext [rows, D] = ZFilter.forward(obs_next rows) (optional: the critic's tail rows, ppo.py:380);
 *   ref_std [rows, A] (row stride ld_ref) = exp(ref_log_var): the std columns of ref_pol (builders.py:126-129);
 *   the packed weight copies of n_pack networks (smx_epoch_pack_f32);
 *   zero_words[0 .. n_zero) = 0 (the control block's stop flag / epoch counter / statistics rows).
 * obs0 / obs_next are row-strided views (ld in floats, e.g. N * D: step 0 of every sub-trajectory). */
typedef struct smx_epoch_prep {
    const float* obs0; int64_t ld_obs0; int64_t rows; int32_t D; int32_t A;
    const float* zmean; const float* zstd;
    float* xn; float* xnT; int64_t ldT;
    const float* ref_sum; const float* ref_sumsq; const float* ref_count; float ref_eps; int32_t ref_filter;
    float* xr;
    const float* obs_next; int64_t ld_next; float* xnext;
    const float* ref_log_var; float* ref_std; int64_t ld_ref;
    int32_t* zero_words; int32_t n_zero; int32_t n_pack;
    smx_epoch_pack_t pack[4];
} smx_epoch_prep_t;
int smx_epoch_prepare_f32(const smx_epoch_prep_t* args, smx_stream_t stream);
struct smx_ppo_losses;
int32_t smx_epoch_blocks(int64_t rows);
int32_t smx_epoch_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_epoch_forward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const struct smx_ppo_losses* loss,
                          smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream);
int smx_epoch_backward_f32(const smx_epoch_job_t* jobs, int32_t njobs, const struct smx_ppo_losses* loss,
                           smx_ppo_ctrl_t* ctrl, int64_t n_total, smx_stream_t stream);
/* smx_epoch_forward_f32 + smx_epoch_backward_f32 of an UPDATING epoch (loss->will_update != 0; surreal/learner/
 * ppo.py:209-248 / 266-309 / 323-353 up to the data gradients) in ONE launch: a workgroup carries its 16 rows through
 * the three layers, the loss and back to dz2 / dz1 while the activations (the ReLU masks) and the loss's gradient
 * terms are still in LDS.  The batch means the adapt loss needs (c_kl = beta + 2 eta max(0, KL - 2 kl_target),
 * ppo.py:272-276) travel INSIDE the launch: every actor workgroup stores its block's KL sum together with a "there"
 * bit as ONE 8-byte device-scope word (kl_slots[block]), multiplies both right-hand sides (g_surr / n, g_kl / n)
 * through the output layer meanwhile, then reads the slots of all blocks (bounded wait: 0.25 s, then ctrl->reserved[1]
 * is raised and the optimiser launch skips its step), adds them in a fixed order and continues with
 * dz2 = (W3^T g_surr + c_kl W3^T g_kl) * relu'(h2) -- the combination is formed one layer later than
 * smx_epoch_backward_f32 forms it (rounding differs in the last bit, the contract is the same).  Clip mode: no
 * gradient depends on the batch, nobody waits.  The epoch's scalars (statistics, log_var's gradient, early-exit flag,
 * step counters) are formed once, by the last workgroup of the grid after its own rows, from the partial rows
 * (device-scope stores; *sync_word counts the actor workgroups whose row is complete).  loss->g_surr / g_kl are not
 * written (the tiles stay in LDS).
 * Jobs: SMX_EPOCH_LOSS_POLICY (at most one) and / or SMX_EPOCH_LOSS_VALUE, fields as for the two calls it
 * replaces.  sync_word: one int32 per launch; kl_slots: smx_epoch_blocks(rows of the policy job) 8-byte words per
 * launch; both zero on entry (the caller clears them once per learn).
 * smx_epoch_fwdbwd_supported: smx_epoch_supported and H2 <= 384 and the larger LDS carve-up fits.
 * Adapt mode needs every actor workgroup of the launch resident at once (one per CU): a launch with more workgroups than
 * the device has CUs runs as smx_epoch_forward_f32 + smx_epoch_backward_f32 (two launches, same results). */
int32_t smx_epoch_fwdbwd_supported(int32_t D, int32_t H1, int32_t H2, int32_t OUT);
int smx_epoch_fwdbwd_f32(const smx_epoch_job_t* jobs, int32_t njobs, const struct smx_ppo_losses* loss,
                         smx_ppo_ctrl_t* ctrl, int64_t n_total, int32_t* sync_word, uint64_t* kl_slots,
                         smx_stream_t stream);
/* A co-tenant for the launch above (diagnostics / tests; no reference counterpart): `blocks` workgroups, each holding
 * one compute unit to itself (more than half of its LDS) for `microseconds` (<= 2 s) without doing work.  The in-launch
 * wait of smx_epoch_fwdbwd_f32 is bounded at 0.25 s: a tenant that keeps its workgroups off the device for longer makes
 * that learn fail loudly (ctrl->reserved[1]); a learner that may share the device uses the two-launch form
 * (session_config.learner.exclusive_device = False). */
int smx_device_occupy(int32_t blocks, int64_t microseconds, smx_stream_t stream);

/* --- acting head (PPOAgent.act, ppo_agent.py:106-154; DiagGauss.sample/maxprob, ppo_net.py:74-91) ---
 * pd[r] = [mean[r, :], exp(log_var) * noise_scale[r]]   (builders.py:127; ppo_agent.py:139:
 *         action_pd[:, A:] *= exp(noise), noise_scale == NULL: 1)
 * actions[r] = clip(eps[r] * std + mean, -1, 1)         (ppo_net.py:80-82, ppo_agent.py:147),
 *              eps == NULL: clip(mean) -- the deterministic evaluation modes.
 * eps is the caller's standard-normal draw (one row per actor); every ld is a row stride in
 * floats, so pd / actions can be slots of a rollout buffer.  pd may be NULL. */
int smx_diaggauss_sample_f32(const float* mean, int64_t ld_mean, const float* log_var,
                             const float* noise_scale, const float* eps, int64_t ld_eps,
                             int64_t rows, int32_t A, float* actions, int64_t ld_act, float* pd,
                             int64_t ld_pd, smx_stream_t stream);
/* The means PPOLearner._optimize reports once per learn, formed on the device so that the whole
 * statistics block needs one read-back: out4 = {mean(log_var) (ppo.py:572), mean_d(running_sum/
 * count), mean_d(running_sumsq/count), mean_d(sqrt(running_sumsq/count - (running_sum/count)^2))}
 * (ppo.py:580-583, z_filter.py:81-107; running_sum == NULL: only out4[0]). */
int smx_ppo_final_stats_f32(const float* log_var, int32_t A, const float* running_sum,
                            const float* running_sumsq, const float* count, int32_t D, float* out4,
                            smx_stream_t stream);
/* stats[e, SMX_VS_*] for e < count from partials [count, nblk, 8] (one launch per learn) */
int smx_value_loss_finalize_f32(const float* partials, int32_t count, int32_t nblk,
                                float* stats, int32_t stats_stride, smx_stream_t stream);

/* --- clip_grad_norm_ + Adam (ppo.py:243-247,304-308,348-352; torch.optim.Adam) ----
 * theta/grads/exp_avg/exp_avg_sq flat [n]; sumsq_partials [npart]: sums of squares of
 * disjoint pieces of `grads` (per-tile values from smx_mlp3_backward_f32 followed by any
 * extra entries the caller appended, e.g. log_var's; or smx_sumsq_partials_f32 of the
 * all-reduced gradient).  which = 0 actor / 1 critic selects lr, max_norm, weight decay and
 * the (already advanced) step counter in ctrl.  Writes the pre-clip total norm to
 * *grad_norm_out.  No-op when ctrl->stop_flag is set and honour_stop != 0. */
int smx_clip_adam_step_f32(float* theta, const float* grads, float* exp_avg,
                           float* exp_avg_sq, int64_t n, const float* sumsq_partials,
                           int32_t npart, const smx_ppo_ctrl_t* ctrl, int32_t which,
                           int32_t honour_stop, float* grad_norm_out, smx_stream_t stream);
/* The same step for the actor's group AND the critic's group in one launch (a lock-step epoch
 * updates both: ppo.py:541-562 touches disjoint optimisers).  `which` is implied: actor = 0,
 * critic = 1. */
typedef struct smx_adam_group {
    float* theta;
    const float* grads;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    const float* sumsq_partials;
    int32_t npart;
    int32_t honour_stop;
    float* grad_norm_out;
    /* optional (both or neither): the group's MLP and its packed copy (smx_epoch_pack_f32 layout); the
     * step then writes every updated weight into the packed copy as well, so the next fused epoch
     * forward needs no separate packing launch */
    const smx_mlp3_t* pack_net;
    float* packed;
} smx_adam_group_t;
/* one group (which: 0 = actor's scalars of the control block, 1 = critic's), with the optional packed copy */
int smx_clip_adam_step_group_f32(const smx_adam_group_t* group, int32_t which, const smx_ppo_ctrl_t* ctrl,
                                 smx_stream_t stream);
int smx_clip_adam_step_pair_f32(const smx_adam_group_t* actor, const smx_adam_group_t* critic,
                                const smx_ppo_ctrl_t* ctrl, smx_stream_t stream);
/* partials[b] = sum of squares of block b's slice of x; returns via *nblk_out the count
 * used (<= max_blocks). */
int32_t smx_sumsq_blocks(int64_t n);
int smx_sumsq_partials_f32(const float* x, int64_t n, float* partials, smx_stream_t stream);

/* --- replay buffers (surreal/replay/fifo_replay.py:27-48, uniform_replay.py:36-47) ---
 * Storage is struct-of-arrays in HBM: one [capacity, width] fp32 table per field.
 * ring insert: table[(cursor + i) % capacity, :] = src[i, :] for i < n */
int smx_ring_insert_f32(float* table, int64_t capacity, int32_t width, int64_t cursor,
                        const float* src, int64_t n, smx_stream_t stream);
/* gather: dst[i, :] = table[idx[i], :]  (uniform sample with injected / generated idx,
 * FIFO pop with idx = (head + i) % capacity) */
int smx_gather_rows_f32(const float* table, int64_t capacity, int32_t width,
                        const int64_t* idx, int64_t n, float* dst, smx_stream_t stream);
/* with-replacement uniform indices in [0, len) from a Philox4x32-10 stream
 * (random.randint(0, len-1) per draw in the reference, uniform_replay.py:44-45) */
int smx_uniform_indices(int64_t* idx, int64_t n, int64_t len, uint64_t seed,
                        uint64_t offset, smx_stream_t stream);
/* The generator behind it by itself -- Philox4x32-10 (Salmon et al., SC'11) on arbitrary counters and keys:
 * ctr_key [n, 6] = {ctr0..3, key0, key1} -> out [n, 4].  The sampler's row i uses ctr = (lo32(offset + i),
 * hi32(offset + i), 0, 0), key = (lo32(seed), hi32(seed)) and idx = mulhi64((out0 << 32) | out1, len).  Exists so that
 * the device code can be checked against Random123's published known-answer vectors (tests/philox_ref.py). */
int smx_philox4x32_10(const uint32_t* ctr_key, int64_t n, uint32_t* out, smx_stream_t stream);
/* UniformReplay.sample (surreal/replay/uniform_replay.py:36-47) over a device-resident replay in ONE launch: the rows
 * idx[i] (or, idx == NULL, the rows smx_uniform_indices(len, seed, offset) would draw -- the same Philox counters) of up
 * to 8 field tables [capacity, row_bytes] -> dst [rows, row_bytes] each.  idx_out (nullable) receives the indices. */
typedef struct {
    const void* table;
    void* dst;
    int64_t row_bytes;
} smx_gather_job_t;
int smx_uniform_gather_multi(const smx_gather_job_t* jobs, int32_t njobs, int64_t capacity, int64_t rows,
                             const int64_t* idx, int64_t len, uint64_t seed, uint64_t offset, int64_t* idx_out,
                             smx_stream_t stream);

/* The same two copies for fields of any element width: a row is `row_bytes` opaque bytes (uint8 camera frames --
 * `pixel_input`, surreal/env/wrapper.py observation specs -- stay uint8 in HBM: a quarter of the table and copy
 * bytes of an fp32 widening).  Moves 16-byte lanes when pitch and addresses allow, else dwords, else bytes. */
int smx_ring_insert_bytes(void* table, int64_t capacity, int64_t row_bytes, int64_t cursor,
                          const void* src, int64_t n, smx_stream_t stream);
int smx_gather_rows_bytes(const void* table, int64_t capacity, int64_t row_bytes, const int64_t* idx,
                          int64_t n, void* dst, smx_stream_t stream);

/* --- sub-trajectory windowing (surreal/env/exp_sender_wrapper.py:209-264) -------
 * From per-actor rollouts laid out [actors, T, width] emit W moving windows of n_step rows:
 *   dst[(a*W + w), j, :] = src[a, start + w*stride + j, :]     0 <= j < n_step
 * The reference emits floor((T - n_step)/stride) + 1 windows per episode of T steps
 * (start = 0); obs_next of window w is the same call with start = n_step, n_step = 1 on a
 * rollout that holds T+1 observations. */
int smx_window_emit_f32(const float* src, int32_t actors, int32_t T, int32_t width,
                        int32_t start, int32_t n_step, int32_t stride, int32_t W, float* dst,
                        smx_stream_t stream);
/* byte-width variant (uint8 frames): rows of `row_bytes` bytes, same index arithmetic */
int smx_window_emit_bytes(const void* src, int32_t actors, int32_t T, int64_t row_bytes, int32_t start,
                          int32_t n_step, int32_t stride, int32_t W, void* dst, smx_stream_t stream);

/* --- "obs stacking" (FrameStackWrapper, surreal/env/wrapper.py:407-472) over a device-resident rollout -----------
 * frames [actors, R, frame_bytes] holds ONE raw camera frame per step (row 0: the frame after reset).  The stacked
 * observation of row s is the last n_stack frames on the channel axis, oldest first; a reset fills the history with
 * the first frame:   stacked(a, s)[i] = frames[a, max(s - (n_stack - 1) + i, first(a, s))],   0 <= i < n_stack,
 * first(a, s) = episode_first ? episode_first[a*R + s] : 0 (the row at which the episode that row s belongs to began).
 * Emitted as W moving windows of n_step rows (smx_window_emit_bytes' arithmetic) in the same gather:
 *   dst[(a*W + w), j, i, :] = stacked(a, start + w*stride + j)[i]      dst: [actors*W, n_step, n_stack, frame_bytes]
 * (W = 1, n_step = 1, start = t: what every actor's policy sees at step t.)  The raw frames are stored once; the
 * n_stack-fold copy of every frame that stacking on the host keeps per step is never made. */
int smx_frame_stack_u8(const void* frames, int32_t actors, int32_t R, int64_t frame_bytes, int32_t n_stack,
                       const int32_t* episode_first, int32_t start, int32_t n_step, int32_t stride, int32_t W,
                       void* dst, smx_stream_t stream);
/* the synthetic environment's camera (no reference counterpart: the reference renders MuJoCo): for every actor a,
 * dst[a*ld_dst + (c, y, x)] = (37 c + 5 y + 11 x + 3 t + (int)(100 |s0[a*ld_s0]|)) % 256, uint8 [C, H, W]. */
int smx_synth_frame_u8(const float* s0, int64_t ld_s0, int32_t n, int32_t C, int32_t H, int32_t W, int32_t t,
                       void* dst, int64_t ld_dst, smx_stream_t stream);

/* --- synthetic vectorised environment step ("batched vectorised env stepping") --------
 * There is no reference counterpart (the reference steps MuJoCo simulators one process per
 * agent, surreal/agent/base.py:244-271); this is the synthetic stand-in BASELINE.json names,
 * stepped for all actors of a GPU in one launch and recorded straight into the device rollout:
 *   obs_roll[a, slot, :] = state[a, :]; obs_roll[a, slot+1, :] = state'[a, :] (if slot+1 < T);
 *   act_roll[a, slot, :] = clip(actions[a, :], -1, 1)   (every roll has T rows per actor)
 *   state'[a, k] = clamp(0.9*state[a,k] + 0.5*act[a, k % A] + 0.01*((37*k) % 17 - 8), -10, 10)
 *   rew_roll[a, slot] = -0.1 * sum_j act[a,j]^2 + 0.05 * state'[a, 0]
 *   done_roll[a, slot] = (t + 1 >= episode_len); on done the state resets to
 *   init_state[a, :] (a fresh episode), mirroring Agent.main_loop's reset.
 * Rolls may be NULL (pure stepping). */
int smx_synth_env_step_f32(float* state, const float* init_state, const float* actions,
                           int32_t n, int32_t D, int32_t A, int32_t t, int32_t episode_len,
                           int32_t slot, int32_t T, float* obs_roll, float* act_roll,
                           float* rew_roll, float* done_roll, smx_stream_t stream);

/* One launch between two policy forwards of a device-resident rollout: the acting head
 * (smx_diaggauss_sample_f32 on `mean`), the environment step above with the sampled action, and the
 * z-filter of the NEXT observation (smx_zfilter_forward_sums_f32) into xn_out -- what the next
 * policy forward reads.  pd_roll [n, T, 2A] receives the action_infos distribution; eps == NULL:
 * deterministic; zsum == NULL: xn_out = the raw next observation; rolls may be NULL. */
typedef struct smx_synth_act_step {
    float* state;
    const float* init_state;
    const float* mean;
    const float* log_var;
    const float* noise_scale;
    const float* eps;
    int64_t ld_mean, ld_eps;
    int32_t n, D, A, t, episode_len, slot, T, reserved;
    float* obs_roll;
    float* act_roll;
    float* rew_roll;
    float* done_roll;
    float* pd_roll;
    const float* zsum;
    const float* zsumsq;
    const float* zcount;
    float zeps, reserved_f;
    float* xn_out;
} smx_synth_act_step_t;
int smx_synth_act_env_step_f32(const smx_synth_act_step_t* args, smx_stream_t stream);
/* The same launch with the policy's output layer folded in (PPOAgent.act's last Linear + Tanh, surreal/model/
 * model_builders/builders.py:126-135): mean = act(h2 . W3^T + b3) is formed per actor inside (A <= 32; args->mean is
 * ignored), so an acting step of a device-resident rollout is three dependent launches instead of four. */
int smx_synth_act_env_step_head_f32(const smx_synth_act_step_t* args, const float* W3, const float* b3,
                                    const float* h2, int64_t ld_h2, int32_t H2, int32_t out_act,
                                    smx_stream_t stream);


/* --- DDPG update pieces (surreal/learner/ddpg.py:244-352, 403-428) --------------------
 * Dense layers reuse smx_linear_f32 / smx_mlp3_*; the critic's "concat action into layer 2"
 * (builders.py:58-84) is expressed with row strides: layer 1 writes into the first c1 columns
 * of a [rows, c1+A] buffer whose last A columns hold the action.
 *   y = rewards + gamma^n * Q'(s', mu'(s')) * (1 - done)   (ddpg.py:279)
 *   dz3 = 2 (Q - y) / rows = d MSELoss / dQ                (ddpg.py:307-308) */
int smx_ddpg_critic_loss_f32(const float* q, const float* q_next_target, const float* rewards,
                             const float* dones, float gamma_n, int64_t rows, float* y,
                             float* dz3, smx_stream_t stream);
/* out = da * (1 - a^2): backward of the actor's final Tanh (builders.py:50) */
int smx_tanh_backward_f32(const float* da, const float* a, int64_t n, float* out,
                          smx_stream_t stream);
int smx_fill_f32(float* x, int64_t n, float value, smx_stream_t stream);
/* LayerNorm over the last dimension, forward and backward (DDPG use_layernorm = True: L.LayerNorm(1) behind every hidden
 * ReLU of ActorNetworkX / CriticNetworkX, model_builders/builders.py:42-48, 65-75; taken as torch.nn.LayerNorm(F):
 * biased variance, eps inside the root, elementwise affine -- torchx's source is not in the reference tree).
 *   forward : y[r, :] = (x[r, :] - mean_r) * rstd_r * gamma + beta; mean / rstd [rows] saved for the backward (may be NULL)
 *   backward: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; relu_mask != 0: dx *= (x > 0) (x is the output
 *             of the ReLU in front of the LayerNorm: dx is then the gradient at that ReLU's input);
 *             dgamma[j] = sum_r dy xhat, dbeta[j] = sum_r dy (overwritten; summed in a fixed order through ws,
 *             smx_layernorm_backward_ws_floats(rows, F) floats).  F <= 1024. */
int smx_layernorm_forward_f32(const float* x, int64_t ldx, int64_t rows, int32_t F, const float* gamma, const float* beta,
                              float eps, float* y, int64_t ldy, float* mean, float* rstd, smx_stream_t stream);
int64_t smx_layernorm_backward_ws_floats(int64_t rows, int32_t F);
int smx_layernorm_backward_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                               const float* rstd, const float* gamma, int64_t rows, int32_t F, int32_t relu_mask, float* dx,
                               int64_t lddx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, smx_stream_t stream);
/* torch.optim.Adam step with optional clip_grad_value_ (Module.clip_grad_value, ddpg.py:309,332);
 * step = 1-based Adam step count; clip_value <= 0 disables clipping. */
int smx_adam_step_f32(float* theta, const float* grads, float* exp_avg, float* exp_avg_sq,
                      int64_t n, double lr, int32_t step, double weight_decay,
                      double clip_value, smx_stream_t stream);
/* The pieces that let ONE captured hipGraph be replayed for every DDPG iteration: the Adam step
 * count and the learning rates live in device memory.
 *   smx_ddpg_critic_loss_step_f32  = smx_ddpg_critic_loss_f32 + (*step_counter += 1): the
 *                                    iteration's step count, read by both Adam launches after it
 *   smx_adam_step_dev_f32          = smx_adam_step_f32 with lr / step read from the device
 *   smx_hard_update_every_f32      target <- source when *step % interval == 0 (the hard target
 *                                    update of ddpg.py:403-409, decided on the device) */
int smx_ddpg_critic_loss_step_f32(const float* q, const float* q_next_target, const float* rewards,
                                  const float* dones, float gamma_n, int64_t rows, float* y,
                                  float* dz3, int32_t* step_counter, smx_stream_t stream);
int smx_adam_step_dev_f32(float* theta, const float* grads, float* exp_avg, float* exp_avg_sq,
                          int64_t n, const float* lr, const int32_t* step, double weight_decay,
                          double clip_value, smx_stream_t stream);
int smx_hard_update_every_f32(float* target, const float* source, int64_t n, const int32_t* step,
                              int32_t interval, smx_stream_t stream);
/* target = target*(1-tau) + source*tau ; tau >= 1 is hard_update (ddpg.py:410-428) */
int smx_soft_update_f32(float* target, const float* source, float tau, int64_t n,
                        smx_stream_t stream);
/* stats[7] = {actor_loss = -mean q_actor, critic_loss = mean (q-y)^2, action_norm = mean
 * ||a||_2, mean rewards, Q_target = mean y, Q_policy = mean q   (ddpg.py:335-342),
 * max |a| (NaN if any action is NaN) for the |actions| <= 1 check of ddpg.py:262-263} */
int smx_ddpg_stats_f32(const float* q, const float* y, const float* rewards,
                       const float* actions, int32_t ld_act, int32_t A, const float* q_actor,
                       int64_t rows, float* stats, smx_stream_t stream);

/* --- one DDPG iteration on ROW BLOCKS (round 5; surreal/learner/ddpg.py:244-352, low-dimensional observations, one critic) ---
 * The layer-by-layer schedule above is ~19 dependent launches of 512-row problems.  Batch rows are independent up to the
 * weight gradients, so a workgroup carries FOUR rows through whole chains (round 6: the v_mfma_f32_4x4x1 loop of the
 * rollout kernel, the chain as a table of layer steps in the kernel arguments; the learner uses it up to 1024 rows per rank):
 *   smx_ddpg_rows_critic_f32  mu'(s') -> Q'(s', mu'(s')) (target networks); Q(s, a); y = r + gamma^n Q' (1 - done) and
 *                             dz3 = 2 (Q - y) / rows (ddpg.py:279, 307-308); *step += 1; the critic's data gradients
 *                             dz2 [rows, c2] and dz1 (first c1 columns of dxcat); mu(s) for the actor phase (h1a, h2a, act)
 *   smx_ddpg_rows_actor_f32   Q(s, mu(s)) through the critic as it is NOW (after its step, which keeps the
 *                             packed copy current) -> q_actor; d(-mean Q)/d(action) through tanh
 *                             -> dz3a; the actor's data gradients dz2a, dz1a (ddpg.py:326-331)
 * The weight gradients (sums over all rows) and the optimiser step are smx_ddpg_rows_wgrad_update_f32 (one rank) or the
 * launches declared above around smx_ddpg_rows_update_f32 (several), on the same row-major buffers.  Weights are read from a copy in MFMA fragment order (`packed`,
 * smx_ddpg_rows_packed_floats floats, 16-byte aligned) which smx_ddpg_rows_pack_f32 refreshes from the row-major
 * parameters: every network (SMX_DDPG_PACK_ALL) or the critic's blocks only.  xcat / dxcat have row stride c1 + A.
 * H1, H2, c1, c2 multiples of 4, A <= 32: smx_ddpg_rows_supported; otherwise SMX_E_UNSUPPORTED. */
typedef struct smx_ddpg_net {          /* nn.Linear layouts: W [out, in] row-major */
    const float *W1, *b1, *W2, *b2, *W3, *b3;
} smx_ddpg_net_t;
typedef struct smx_ddpg_rows {
    int64_t rows;
    int32_t D, A, H1, H2, c1, c2;      /* actor D -> H1 -> H2 -> A (tanh); critic D -> c1, [c1 | A] -> c2 -> 1 */
    smx_ddpg_net_t actor, critic, target_actor, target_critic;
    float* packed;
    const float *x, *x_next, *actions, *rewards, *dones;       /* [rows, D] x 2, [rows, A], [rows] x 2 */
    float gamma_n;
    float *xcat, *h2c, *q, *q_next, *y, *dz3, *dz2, *dxcat;    /* critic phase, out */
    float *h1a, *h2a, *act;                                    /* critic phase out, actor phase in */
    float *q_actor, *dz3a, *dz2a, *dz1a;                       /* actor phase, out */
    int32_t* step;                                             /* device Adam step counter (may be NULL) */
} smx_ddpg_rows_t;
enum { SMX_DDPG_PACK_ALL = 0, SMX_DDPG_PACK_CRITIC = 1 };
int32_t smx_ddpg_rows_supported(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2);   /* on 4-row blocks */
/* ... for a batch of `rows` (the same answer for every row count since the 16-row kernels of round 5 are gone) */
int32_t smx_ddpg_rows_supported_at(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2, int64_t rows);
int64_t smx_ddpg_rows_packed_floats(int32_t D, int32_t A, int32_t H1, int32_t H2, int32_t c1, int32_t c2);
int smx_ddpg_rows_pack_f32(const smx_ddpg_rows_t* args, int32_t which, smx_stream_t stream);
int smx_ddpg_rows_critic_f32(const smx_ddpg_rows_t* args, smx_stream_t stream);
int smx_ddpg_rows_actor_f32(const smx_ddpg_rows_t* args, smx_stream_t stream);

/* One optimiser group's step of the row schedule in ONE launch (round 6): Adam exactly as smx_adam_step_dev_f32
 * (torch.optim.Adam after clip_grad_value_: ddpg.py:310-311, 332-333), then the group's target network -- soft
 * (interval == 0: target = target (1 - tau) + theta tau) or hard every `interval` iterations of *step (ddpg.py:389-428),
 * target == NULL: none -- and the fragment-order copies of both inside args->packed, element by element: no
 * smx_ddpg_rows_pack_f32 is needed between iterations as long as nothing else writes the parameters.  theta / target are
 * the group's parameter buffers ([n], holding the W1, W2, W3 that args names for the group at the same offsets). */
typedef struct smx_ddpg_update {
    float* theta;
    const float* grads;
    float *exp_avg, *exp_avg_sq;
    float* target;
    int64_t n;
    const float* lr;                   /* device */
    const int32_t* step;               /* device: Adam's step count of this iteration (also decides a hard update) */
    float weight_decay, clip_value, tau;
    int32_t interval;
    float* stats;                      /* smx_ddpg_rows_wgrad_update_f32 only, may be NULL: [7] as smx_ddpg_stats_f32 writes them,
                                          from args' q, y, rewards, actions, q_actor -- one more workgroup of the launch */
    float* stats_host;                 /* with stats, may be NULL: device-accessible HOST memory [2][8]; the same seven words also
                                          go to slot (*step & 1) -- the caller reads them after the launch without a copy */
} smx_ddpg_update_t;
enum { SMX_DDPG_GROUP_ACTOR = 0, SMX_DDPG_GROUP_CRITIC = 1 };
int smx_ddpg_rows_update_f32(const smx_ddpg_rows_t* args, int32_t group, const smx_ddpg_update_t* update,
                             smx_stream_t stream);
/* The same step with the group's weight gradients formed in the SAME launch (one rank: nothing to exchange between them):
 * dW = dz^T x and db = the column sums of dz over args->rows rows, from the buffers the chain launches wrote (critic:
 * dxcat / dz2 / dz3 against x / xcat / h2c; actor: dz1a / dz2a / dz3a against x / h1a / h2a), written to update->grads
 * (laid out like theta) and stepped while still in the accumulators.  update->n must be exactly the three layers'
 * weights and biases. */
int smx_ddpg_rows_wgrad_update_f32(const smx_ddpg_rows_t* args, int32_t group, const smx_ddpg_update_t* update,
                                   smx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LSTM stem (surreal/model/ppo_net.py:143-152: nn.LSTM(in, rnn_hidden, 1, batch_first=True) in
 * front of the actor / critic MLPs; forward at :277-279, :307-309, single step at :338-349).
 * torch.nn.LSTM layouts: W_ih [4H, D], W_hh [4H, H], b_ih [4H], b_hh [4H], gate order i, f, g, o.
 * H must be a multiple of 4 and <= 384; B*T*4H < 2^31.
 * ------------------------------------------------------------------------------------------- */
typedef struct smx_lstm {
    const float* W_ih;
    const float* W_hh;
    const float* b_ih;
    const float* b_hh;
    int32_t D, H;
} smx_lstm_t;

/* number of parameters, laid out [W_ih | W_hh | b_ih | b_hh] (also the layout of `grads` below) */
int64_t smx_lstm_param_count(int32_t D, int32_t H);

/* out[b, t, :] = h_t for x [B, T, D] starting from (h0, c0) [B, H] (NULL = zeros).  Saved for
 * the backward pass: gates [B, T, 4H] (activated i, f, g, o), cs [B, T, H] (c_t), hprev
 * [B, T, H] (h_{t-1}, may be NULL for inference).  hN / cN [B, H] (optional) receive the final
 * state (forward_actor_expose_cells).  A non-zero *stop_flag skips the whole call. */
int smx_lstm_forward_f32(const smx_lstm_t* net, const float* x, int64_t B, int32_t T,
                         const float* h0, const float* c0, float* gates, float* out, float* cs,
                         float* hprev, float* hN, float* cN, const int32_t* stop_flag,
                         smx_stream_t stream);

/* Back-propagation through time given dout [B, T, H] = dLoss/dh_t from the layers above
 * (h0, c0 are constants: ppo.py:511-515 detaches them).  dgates [B, T, 4H] is workspace (it may
 * alias `gates`).  grads receives dW_ih, dW_hh, db_ih, db_hh (overwritten, not accumulated).
 * ws / ws_floats: optional split-K workspace for the two weight-gradient GEMMs over B*T rows
 * (smx_lstm_backward_ws_floats; NULL = unsplit). */
int64_t smx_lstm_backward_ws_floats(int32_t D, int32_t H, int64_t B, int32_t T);
int smx_lstm_backward_f32(const smx_lstm_t* net, const float* x, int64_t B, int32_t T,
                          const float* c0, const float* gates, const float* cs,
                          const float* hprev, const float* dout, float* dgates, float* grads,
                          const int32_t* stop_flag, float* ws, int64_t ws_floats,
                          smx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * CNN stem data movement (surreal/model/model_builders/builders.py:8-33 CNNStemNetwork;
 * ppo_net.py:268-273, 368-375 `_scale_image`).  The convolutions themselves are smx_linear_f32 /
 * smx_linear_wgrad_f32 over patch rows.
 * ------------------------------------------------------------------------------------------- */
/* cols[(f*Ho + oy)*Wo + ox, c*kh*kw + i*kw + j] = src(f, c, oy*stride + i, ox*stride + j)
 * [/ scale_div when scale_div != 0: x / 255.0 of _scale_image].  src: uint8 or fp32 frames
 * [F, C, Hin, Win] (channel_last = 0), or an fp32 activation [F, Hin*Win, C] (channel_last = 1).
 * No padding (torch default); Ho = (Hin - kh)/stride + 1.  The k order is torch's
 * Conv2d.weight.view(out, -1) order. */
int smx_im2col_f32(const void* src, int32_t src_is_u8, int32_t channel_last, int64_t F, int32_t C,
                   int32_t Hin, int32_t Win, int32_t kh, int32_t kw, int32_t stride,
                   float scale_div, float* cols, smx_stream_t stream);
/* First convolution over uint8 frames as an implicit GEMM (no patch matrix): y[(f, oy, ox), o] = relu(b[o] +
 * sum_{c,i,j} W[o][c][i][j] * float(frames[f][c][oy*stride + i][ox*stride + j]) / 255) -- Conv2d + ReLU of
 * surreal/model/model_builders/builders.py:8-33 on `x / 255` (ppo_net.py:268-275), the same numbers as
 * smx_im2col_f32 (scale_div 255) + smx_linear_f32 (ReLU).  y is channel-last [F*Ho*Wo, cout].
 * SMX_E_UNSUPPORTED unless cout <= 16, k % 4 == 0, Win % 4 == 0, stride % 4 == 0, C*k*k in {64, 128, 192, 256}.
 * stop_flag (device, may be NULL): non-zero turns the launch into a no-op (the KL early exit). */
int smx_conv_u8_forward_f32(const void* frames, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                            int32_t stride, const float* W, const float* bias, int32_t cout, float* y,
                            const int32_t* stop_flag, smx_stream_t stream);
/* The same layer's weight gradient, again from the uint8 frames (no patch matrix): dW[o][(c, i, j)] = sum over
 * (f, oy, ox) of dy[(f, oy, ox), o] * float(frames[f][c][oy*stride + i][ox*stride + j]) / 255, db[o] = the column
 * sums of dy -- what autograd gives Conv2d.weight / .bias (ppo.py:243-247 backward through builders.py:8-33); equal
 * to smx_im2col_f32 + smx_linear_wgrad_splitk_f32 up to summation order.  dy is [F*Ho*Wo, cout] row-major; ws holds
 * smx_conv_u8_wgrad_ws_floats(cout, C*k*k) floats (per-workgroup partials, added in a fixed order by a second launch).
 * Same shape limits as smx_conv_u8_forward_f32. */
int64_t smx_conv_u8_wgrad_ws_floats(int32_t cout, int32_t K);
int smx_conv_u8_wgrad_f32(const void* frames, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                          int32_t stride, const float* dy, int32_t cout, float* dW, float* db, float* ws,
                          int64_t ws_floats, const int32_t* stop_flag, smx_stream_t stream);
/* A convolution over an fp32 CHANNEL-LAST source of 16 channels ([F, Hin*Win, 16]: the first convolution's output) as
 * implicit GEMMs, forward (+ bias + ReLU) and weight gradient -- the second Conv2d of builders.py:8-33 and what autograd
 * gives its weight / bias; W and dW in torch's [cout][16][k][k] order; y / dy are [F*Ho*Wo, cout].  Equal to
 * smx_im2col_f32 (channel_last = 1) + smx_linear_f32 / smx_linear_wgrad_splitk_f32 up to summation order.
 * SMX_E_UNSUPPORTED unless C == 16, cout <= 32, k in {2, 3, 4}.  ws: smx_conv_cl_wgrad_ws_floats(cout, k) floats. */
int smx_conv_cl_forward_f32(const float* src, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                            int32_t stride, const float* W, const float* bias, int32_t cout, float* y,
                            const int32_t* stop_flag, smx_stream_t stream);
int64_t smx_conv_cl_wgrad_ws_floats(int32_t cout, int32_t k);
int smx_conv_cl_wgrad_f32(const float* src, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                          int32_t stride, const float* dy, int32_t cout, float* dW, float* db, float* ws,
                          int64_t ws_floats, const int32_t* stop_flag, smx_stream_t stream);
/* ... and its data gradient, dx [F, Hin*Win, 16] (channel-last) = relu'(relu_of) * conv_transpose(dy, W), without the
 * intermediate dcols matrix: what smx_linear_f32 (dy . W) + smx_col2im_f32 give, up to summation order.
 * SMX_E_UNSUPPORTED unless C == 16, cout <= 32 and k == 2 * stride. */
int smx_conv_cl_dgrad_f32(const float* dy, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t k,
                          int32_t stride, const float* W, int32_t cout, const float* relu_of, float* dx,
                          const int32_t* stop_flag, smx_stream_t stream);
/* data gradient of the convolution above: dx [F, Hin*Win, C] (channel-last) gathers dcols
 * [F*Ho*Wo, C*kh*kw]; relu_of (optional, same shape as dx): dx *= (relu_of > 0). */
int smx_col2im_f32(const float* dcols, int64_t F, int32_t C, int32_t Hin, int32_t Win, int32_t kh,
                   int32_t kw, int32_t stride, const float* relu_of, float* dx,
                   smx_stream_t stream);
/* Flatten order of the Linear after the convolutions: to_channel_last != 0:
 * out[o, p*C + c] = in[o, c*P + p], else the inverse (used for the weight and for its gradient). */
int smx_flatten_order_f32(const float* in, int32_t O, int32_t C, int32_t P,
                          int32_t to_channel_last, float* out, smx_stream_t stream);

/* A whole device-resident rollout in ONE launch: `steps` iterations of [z-filter -> policy MLP -> DiagGauss sample ->
 * clip -> synthetic env step -> record] for all n actors (the per-step loop of surreal/agent/base.py:244-271 with
 * PPOAgent.act, surreal/agent/ppo_agent.py:106-154, and the recording of env/exp_sender_wrapper.py:153-264).  A
 * workgroup owns 4 actors for the whole rollout (8 from 1025 actors on, 16 beyond 2048: the smallest block whose grid fits
 * the CUs once); the policy layers run on FP32 MFMA (v_mfma_f32_4x4x1; 16x16x4 for 16-actor blocks) from `packed`
 * (smx_epoch_pack_f32 of `net`): the means equal smx_epoch_forward_f32's to fp32 rounding of the layer sums (another
 * summation order), 4- and 8-actor blocks bit for bit each other's; sampling, dynamics, recording and the z-filter use the
 * expressions of smx_synth_act_env_step_f32.
 * eps [steps, n, A] standard normals (NULL: deterministic); zsum/zsumsq/zcount: the z-filter's running sums (NULL:
 * raw observations); rolls [n, rows_per_actor, .] (any may be NULL); state [n, D] is read at the start and left at
 * the state after the last step; t: the episode clock at the first step.
 * obs_last [n, D] (nullable): where the observation AFTER row rows_per_actor - 1 goes.  With rows_per_actor = steps + 1
 * (a rollout table) that observation is the table's last row and obs_last is not used; with rows_per_actor = steps the
 * tables have exactly the replay's layout -- obs [n, n_step, D] + obs_next [n, 1, D] = obs_last -- and the rollout is
 * recorded STRAIGHT INTO the FIFO's slots when stride == n_step (env/exp_sender_wrapper.py:209-228: a window then is
 * the rollout; no window cut, no insert copy). */
typedef struct smx_synth_rollout {
    const smx_mlp3_t* net;
    const float* packed;
    int32_t out_act, n;
    const float* log_var;
    const float* noise_scale;
    const float* eps;
    const float* zsum;
    const float* zsumsq;
    const float* zcount;
    float zeps;
    int32_t t, episode_len, steps, rows_per_actor, slot;
    float* state;
    const float* init_state;
    float* obs_roll;
    float* act_roll;
    float* rew_roll;
    float* done_roll;
    float* pd_roll;
    float* obs_last;
} smx_synth_rollout_t;
int32_t smx_synth_rollout_supported(int32_t D, int32_t H1, int32_t H2, int32_t A);
int smx_synth_rollout_f32(const smx_synth_rollout_t* args, smx_stream_t stream);

/* ---------------------------------------------------------------------------
 * Data-parallel exchange between the learner ranks of one node over IPC-mapped peer buffers (xGMI loads): the
 * collectives N sharded learners need to equal the single reference learner -- the per-epoch
 * [gradients | loss partial rows] sum of surreal/learner/ppo.py:541-562 run on shards, the advantage moments
 * (ppo.py:413-416), the end-of-learn statistics -- as ONE kernel each inside the learner's hipGraph instead of an
 * eager RCCL call between graph segments (csrc/smx_xchg.hip has the protocol).  The reference has no counterpart (one
 * learner process); the host-side fallback is torch.distributed (RCCL).
 *
 * Set-up, once per learner workspace: every rank  smx_xchg_alloc()s a buffer of smx_xchg_bytes(capacity, world),
 * smx_xchg_export()s its 64-byte handle, the handles travel through the process group, every rank smx_xchg_open()s
 * its peers' handles and fills smx_xchg_t.peer[] (peer[rank] = its own buffer), then a barrier.  These are the only
 * entry points of the library that allocate or synchronise.
 * Exchanges: issued by every rank in the same order on ONE stream per exchange context; the sequence number lives in
 * the buffer, so a captured graph replays.  err (nullable): a device int32 into which a timed-out wait ORs
 * 0x100 | phase << 4 | peer (the caller reads it back with its statistics); after an error no wait blocks any more.
 * Results are bit-identical on all ranks (rank c alone reduces chunk c, summing the ranks in rank order). */
#define SMX_XCHG_MAX_RANKS 8
#define SMX_XCHG_HANDLE_BYTES 64
typedef struct {
    int32_t world, rank;
    int64_t capacity;                      /* floats one exchange may carry */
    void* peer[SMX_XCHG_MAX_RANKS];        /* every rank's buffer as mapped in THIS process */
} smx_xchg_t;
int64_t smx_xchg_bytes(int64_t capacity_floats, int32_t world);
/* kind (nullable) <- 0 uncached, 1 fine-grained, 2 plain device memory (what the runtime granted); timeout_s: bound of
 * every in-kernel wait (<= 0: 2 s).  Zero-fills and synchronises `stream`. */
int smx_xchg_alloc(int64_t bytes, double timeout_s, void** ptr, int32_t* kind, smx_stream_t stream);
int smx_xchg_free(void* ptr);
int smx_xchg_export(void* ptr, void* handle64);
int smx_xchg_open(const void* handle64, void** ptr);
int smx_xchg_close(void* ptr);
/* out[0, n) = sum over ranks of in[0, n); in == out allowed; 16-byte aligned; n <= capacity */
int smx_xchg_allreduce_f32(const smx_xchg_t* x, const float* in, float* out, int64_t n, int32_t* err,
                           smx_stream_t stream);
/* out [world, n_per_rank] <- every rank's in[0, n_per_rank); world * n_per_rank <= capacity */
int smx_xchg_allgather_f32(const smx_xchg_t* x, const float* in, int64_t n_per_rank, float* out, int32_t* err,
                           smx_stream_t stream);
/* seq_err_dev[2] (device) <- {exchanges completed, error word} of this rank's buffer */
int smx_xchg_status(const smx_xchg_t* x, uint32_t* seq_err_dev, smx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SURREAL_AMD_H */

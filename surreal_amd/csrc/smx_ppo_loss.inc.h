// Device-side pieces of the PPO losses shared by smx_ppo.hip (the layered epoch kernels) and
// smx_epoch.hip (the fused row-block epoch kernels).  Included inside an anonymous namespace.
// Reference: surreal/model/ppo_net.py:29-72, surreal/learner/ppo.py:194-285, 553-557.
#pragma once

// barrier between phases that exchange data through LDS only.  The including file may define it as a
// bare `s_waitcnt lgkmcnt(0); s_barrier`: __syncthreads() also waits for every outstanding GLOBAL
// store of the wave (vmcnt(0)), a ~1 us round trip the fused epoch kernels pay at each of their phases.
#ifndef SMX_LDS_BARRIER
#define SMX_LDS_BARRIER() __syncthreads()
#endif

// rows per workgroup of the policy loss: 16 -> 64 workgroups for the 1024-row epochs (the kernel is
// a chain of three short phases; with 64 rows it ran on 16 CUs and took 11 us, most of it waiting)
constexpr int LOSS_ROWS_PER_BLOCK = 16;
constexpr int MAX_A = 32;

// LDS scratch (floats) of policy_loss_body for A action dimensions
__host__ __device__ constexpr int loss_scratch_floats(int A) { return LOSS_ROWS_PER_BLOCK * (8 * A + 1) + 2 * MAX_A; }
// What policy_loss_body leaves in its scratch for a caller that carries on in the same workgroup (the fused
// forward + backward epoch kernel): per-element d ll / d z3 and d KL / d z3 [R, A], per-row d loss / d ll [R] --
// g_surr[r][a] = row_dll[r] * elem_dmu[r * A + a], g_kl[r][a] = elem_dkl[r * A + a].
__device__ __forceinline__ const float* loss_elem_dmu(const float* sm, int A) { return sm + 5 * LOSS_ROWS_PER_BLOCK * A; }
__device__ __forceinline__ const float* loss_elem_dkl(const float* sm, int A) { return sm + 6 * LOSS_ROWS_PER_BLOCK * A; }
__device__ __forceinline__ const float* loss_row_dll(const float* sm, int A) { return sm + 8 * LOSS_ROWS_PER_BLOCK * A; }
// SMX_LDS_BARRIER()s inside policy_loss_body: wavefronts of the workgroup that do not run the body must execute as many
constexpr int POLICY_LOSS_BARRIERS = 3;

__device__ __forceinline__ float clamp_min_nan(float x, float lo) {
    return (x == x) ? fmaxf(x, lo) : x;  // torch.clamp(min=) keeps NaN
}

// a block's partial sums: plain stores, or (COH) device-scope write-through stores -- for a consumer in ANOTHER workgroup
// of the SAME launch (the fused forward + backward epoch kernel), which reads them with device-scope loads: no
// cache-wide write-back / invalidate (an agent-scope fence on gfx950 is buffer_wbl2 / buffer_inv of the whole L2,
// measured at ~4 us per side with the launch's activations dirty in it)
template <bool COH>
__device__ __forceinline__ void partial_store(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// 64 rows per block, 256 threads, three phases:
//   1. all waves move the block's rows of mean / actions / behave / ref HBM -> LDS with coalesced
//      loads and compute the per-ELEMENT terms (one (row, a) pair per thread-iteration): the
//      transcendentals (log, exp, divisions) are spread over 256 lanes instead of being
//      serialised 17-deep in one lane per row;
//   2. one lane per row reduces its A terms, forms the likelihoods / ratio / clip decision and
//      the per-row gradient scale, and the wave reduces the block partial sums;
//   3. all waves form the two gradient tiles (element-parallel), write them back coalesced, and
//      A lanes reduce the log_var gradient partials over the block's rows.
template <bool COH = false>
__device__ __forceinline__ void policy_loss_body(
    const int blk, float* sm,
    int mode, const float* __restrict__ mean_blk, const int ld_mean, const float* __restrict__ log_var,
    const float* __restrict__ g_actions, int ld_act, const float* __restrict__ g_behave, int ld_beh,
    const float* __restrict__ g_ref, int ld_ref, const float* __restrict__ adv, long rows, int A,
    const smx_ppo_ctrl_t* __restrict__ ctrl, float* __restrict__ g_surr, float* __restrict__ g_kl,
    float* __restrict__ partials, const float gscale = 1.0f, const bool scaled = false,
    float* __restrict__ g_surr_t = nullptr, float* __restrict__ g_kl_t = nullptr, const long ld_t = 0,
    const long in_row0 = 0 /* first row held by g_actions / g_behave / g_ref / adv (a staged block: row0) */,
    const int ld_adv = 1,
    unsigned long long* __restrict__ kl_slot = nullptr /* (COH) the block's KL sum | 1 << 32, ONE 8-byte device-scope
    store as soon as it is known: value and "it is there" travel together (see epoch_fb_kernel) */) {
    const int R = LOSS_ROWS_PER_BLOCK;
    float* e_z2 = sm;              // ((a - mu)/sig)^2                    [R, A]
    float* e_zb2 = e_z2 + R * A;   // ((a - mb)/sb)^2
    float* e_lsb = e_zb2 + R * A;  // log sb
    float* e_kl = e_lsb + R * A;   // log(sig/sr) + (sr^2+(mr-mu)^2)/(2 sig^2)
    float* e_klb = e_kl + R * A;   // log(sb/sr) + (sr^2+(mr-mb)^2)/(2 sb^2)
    float* e_dmu = e_klb + R * A;  // ((a - mu)/sig^2) * (1 - mu^2)        d ll / d z3
    float* e_dkl = e_dmu + R * A;  // ((mu - mr)/sig^2) * (1 - mu^2)       d KL / d z3
    float* e_gk = e_dkl + R * A;   // 1 - (sr^2+(mr-mu)^2)/sig^2          d KL / d log_var
    float* r_dll = e_gk + R * A;   // per-row d(loss_r)/d(ll)             [R]
    float* s_sig = r_dll + R;      // exp(log_var)  (builders.py:127)       [A]
    float* s_lsig = s_sig + MAX_A; // log(exp(log_var)): std0.log()  (ppo_net.py:40)
    const long row0 = (long)blk * R;
    long nrows = rows - row0;
    if (nrows > R) nrows = R;
    const int tid = threadIdx.x;
    const int stride = 8 + 2 * A;
    float* P = partials + (size_t)blk * stride;

    // ---- phase 0: the policy's std and its log, once per block (read 17-deep from global memory by
    // the one-lane-per-row phase they were a chain of dependent cache round trips) --------------------
    if (tid < A) {
        const float sg = expf(log_var[tid]);
        s_sig[tid] = sg;
        s_lsig[tid] = logf(sg);
    }
    SMX_LDS_BARRIER();
    // ---- phase 1: element-parallel terms; a thread's elements i and i + 256 go through the chain of LDS reads,
    // divisions and logarithms TOGETHER (16 x 17 = 272 elements on 256 threads: run one after the other, the sixteen
    // elements of the second pass cost as much as the first 256) ------------------------------------------------
    const int n_el = (int)nrows * A;
    for (int i0 = tid; i0 < n_el; i0 += 512) {
        float o_z2[2], o_zb2[2], o_lsb[2], o_kl[2], o_klb[2], o_dmu[2], o_dkl[2], o_gk[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = min(i0 + 256 * u, n_el - 1);
            const int rr = i / A, a = i - rr * A;
            const long gr = row0 + rr - in_row0;
            const float sig = s_sig[a];
            const float mu = mean_blk[rr * ld_mean + a];      // the block's rows of tanh(z3): global or LDS
            const float ac = g_actions[gr * ld_act + a];
            const float mb = g_behave[gr * ld_beh + a], sb = g_behave[gr * ld_beh + A + a];
            const float mr = g_ref[gr * ld_ref + a], sr = g_ref[gr * ld_ref + A + a];
            const float z = (ac - mu) / sig;                         // ppo_net.py:39
            const float zb = (ac - mb) / sb;
            const float s2 = sig * sig;
            const float dt = 1.0f - mu * mu;                         // tanh'
            const float num = sr * sr + (mr - mu) * (mr - mu);
            o_z2[u] = z * z;
            o_zb2[u] = zb * zb;
            o_lsb[u] = logf(sb);
            o_kl[u] = logf(sig / sr) + num / (2.0f * s2);            // ppo_net.py:61-62
            o_klb[u] = logf(sb / sr) + (sr * sr + (mr - mb) * (mr - mb)) / (2.0f * (sb * sb));
            o_dmu[u] = ((ac - mu) / s2) * dt;
            o_dkl[u] = ((mu - mr) / s2) * dt;
            o_gk[u] = 1.0f - num / s2;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 256 * u;
            if (i < n_el) {
                e_z2[i] = o_z2[u]; e_zb2[i] = o_zb2[u]; e_lsb[i] = o_lsb[u]; e_kl[i] = o_kl[u];
                e_klb[i] = o_klb[u]; e_dmu[i] = o_dmu[u]; e_dkl[i] = o_dkl[u]; e_gk[i] = o_gk[u];
            }
        }
    }
    SMX_LDS_BARRIER();

    // ---- phase 2: FOUR lanes per row (lane 4 r + q takes a = q, q + 4, ...; the quad's partial sums meet in the
    // fixed order (q0 + q1) + (q2 + q3) through two butterfly steps, the same bits in all four lanes): the A-deep
    // chain of dependent LDS reads of one lane per row was the longest single piece of the loss -------------------
    if (tid < 64) {                      // one wave = 16 rows; rows >= nrows only feed zeros to the sums
        const int r = tid >> 2, q = tid & 3;
        const bool ok = r < nrows;
        const float c_ll = (float)(0.5 * 1.8378770664093453 /* log(2 pi) */ * (double)A);
        const float half_d = (float)(0.5 * (double)A);
        float s1 = 0.f, s2 = 0.f, sb1 = 0.f, sb2 = 0.f, klr = 0.f, klbr = 0.f;
        if (ok) {
            for (int a = q; a < A; a += 4) {
                s1 += e_z2[r * A + a];
                s2 += s_lsig[a];
                sb1 += e_zb2[r * A + a];
                sb2 += e_lsb[r * A + a];
                klr += e_kl[r * A + a];
                klbr += e_klb[r * A + a];
            }
        }
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
            sb1 += __shfl_xor(sb1, off, 64);
            sb2 += __shfl_xor(sb2, off, 64);
            klr += __shfl_xor(klr, off, 64);
            klbr += __shfl_xor(klbr, off, 64);
        }
        const float ll = ((-0.5f * s1) - c_ll) - s2;
        const float llb = ((-0.5f * sb1) - c_ll) - sb2;
        const float el = expf(ll);
        const float Ll = clamp_min_nan(el, 1e-5f);               // ppo_net.py:46
        const float Lb = clamp_min_nan(expf(llb), 1e-5f);
        const float kl = klr - half_d;
        const float klb = klbr - half_d;
        const float ad = ok ? adv[(row0 + r - in_row0) * ld_adv] : 0.f;        // (R = 16 rows: r = tid >> 2 < R)
        float surr, loss_r, dLl;  // dLl = d(loss_r)/d(L_learn)
        if (mode == SMX_PPO_CLIP) {
            const float eps = ctrl->clip_eps;
            const float lo = (float)(1.0 - (double)eps), hi = (float)(1.0 + (double)eps);
            const float ratio = Ll / Lb;                                    // ppo.py:212
            float cr = ratio;
            if (cr == cr) cr = fminf(fmaxf(cr, lo), hi);                    // ppo.py:213
            surr = -ratio * ad;                                             // ppo.py:215
            const float cs = -cr * ad;                                      // ppo.py:216
            loss_r = (surr >= cs) ? surr : cs;                              // ppo.py:217
            // max() routes the gradient to the larger entry; the clamped one has zero slope
            // outside [lo, hi] and equals the unclamped one inside.
            dLl = (surr >= cs) ? (-ad / Lb) : 0.f;
        } else {
            const float Lbc = clamp_min_nan(Lb, 1e-2f);                     // ppo.py:271
            surr = -(ad * (Ll / Lbc));
            loss_r = surr;
            dLl = -ad / Lbc;
        }
        // d(loss_r)/d(ll): clamp(min=1e-5) passes the gradient where exp(ll) >= 1e-5
        const float dll = (el >= 1e-5f) ? dLl * el : 0.f;
        if (q == 0) r_dll[r] = ok ? dll : 0.f;
        const float isw = Ll / (Lb + 1e-4f);                                // ppo.py:574
        const bool one = ok && q == 0;                                      // a row enters the block sums once
        const float v0 = smx_wave_sum(one ? surr : 0.f);
        const float v1 = smx_wave_sum(one ? loss_r : 0.f);
        const float v2 = smx_wave_sum(one ? kl : 0.f);
        const float v3 = smx_wave_sum(one ? Lb : 0.f);
        const float v4 = smx_wave_sum(one ? isw : 0.f);
        const float v5 = smx_wave_sum(one ? klb : 0.f);
        if (tid == 0) {
            if (COH && kl_slot)
                __hip_atomic_store(kl_slot + blk, (unsigned long long)__float_as_uint(v2) | (1ull << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            partial_store<COH>(P + 0, v0); partial_store<COH>(P + 1, v1); partial_store<COH>(P + 2, v2);
            partial_store<COH>(P + 3, v3); partial_store<COH>(P + 4, v4); partial_store<COH>(P + 5, v5);
            partial_store<COH>(P + 6, 0.f); partial_store<COH>(P + 7, 0.f);
        }
    }
    SMX_LDS_BARRIER();

    // ---- phase 3: gradient tiles + log_var gradient partials -----------------------------
    // (g_surr == nullptr: the caller takes the tiles from the scratch -- loss_elem_dmu / _dkl / loss_row_dll)
    if (g_surr) {
        for (int i = tid; i < (int)nrows * A; i += 256) {
            const int rr = i / A;
            float gs = r_dll[rr] * e_dmu[i], gk = e_dkl[i];
            if (scaled) { gs *= gscale; gk *= gscale; }       // data-parallel epochs: already / n_total
            g_surr[row0 * A + i] = gs;
            g_kl[row0 * A + i] = gk;
            if (g_surr_t) {
                const long at = (long)(i - rr * A) * ld_t + row0 + rr;
                g_surr_t[at] = gs;
                if (g_kl_t) g_kl_t[at] = gk;
            }
        }
    }
    if (tid < 4 * A) {
        // d ll/d log_var_a = z^2 - 1 ; d KL/d log_var_a = 1 - (sr^2+(mr-mu)^2)/sig^2.  Four lanes per column (lane
        // 4 a + q takes rows q, q + 4, ...), the quad's partial sums meet in the order (q0 + q1) + (q2 + q3)
        const int a = tid >> 2, q = tid & 3;
        float gs = 0.f, gk = 0.f;
        for (int rr = q; rr < (int)nrows; rr += 4) {
            gs += r_dll[rr] * (e_z2[rr * A + a] - 1.0f);
            gk += e_gk[rr * A + a];
        }
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            gs += __shfl_xor(gs, off, 64);
            gk += __shfl_xor(gk, off, 64);
        }
        if (q == 0) {
            partial_store<COH>(P + 8 + a, gs);
            partial_store<COH>(P + 8 + A + a, gk);
        }
    }
}

// batch sums S[0 .. 8 + 2A) of the block partial rows: staged through LDS with coalesced loads and
// added in row order (the order, hence the result, is the same in every workgroup)
constexpr int FIN_CH = 64;                                   // partial rows staged per pass
// 256 threads.  Column c of the partial rows is summed by Q = 256 / stride threads, thread (q, c) taking the rows
// q, q + Q, ... of every staged chunk; the Q chains of a column meet in the fixed order ((s0 + s1) + s2) + ... -- the same
// bits on every run and in every workgroup.  (One chain per column was 7936 dependent additions at the LSTM policy's
// 127 k rows: 197 us per launch, every workgroup of the finalize walking all of them.)
__device__ __forceinline__ void reduce_row_partials(const float* __restrict__ partials, int nblk,
                                                    int stride, float* S, float* buf) {
    const int Q = 256 / stride;                       // stride <= 8 + 2 * MAX_A = 72: at least 3 chains
    const int q = (int)threadIdx.x / stride, c = (int)threadIdx.x - q * stride;
    float t = 0.f;
    for (int b0 = 0; b0 < nblk; b0 += FIN_CH) {
        const int nb = min(FIN_CH, nblk - b0);
        const int cnt = nb * stride;
        const float* src = partials + (size_t)b0 * stride;
        // the chunk's words are requested in batches of eight per thread (a load -> LDS store loop
        // with a run-time trip count is a chain of dependent round trips)
        for (int i0 = threadIdx.x; i0 < cnt; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + 256 * u, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < cnt) buf[i0 + 256 * u] = v[u];
        }
        SMX_LDS_BARRIER();
        if (q < Q)
            for (int b = q; b < nb; b += Q) t += buf[b * stride + c];
        SMX_LDS_BARRIER();
    }
    if (q < Q) buf[q * stride + c] = t;
    SMX_LDS_BARRIER();
    if ((int)threadIdx.x < stride) {
        float s = buf[threadIdx.x];
        for (int k = 1; k < Q; ++k) s += buf[k * stride + threadIdx.x];
        S[threadIdx.x] = s;
    }
    SMX_LDS_BARRIER();
}

// loss and the coefficient of the KL gradient from the batch sums (ppo.py:217 / 272-276)
__device__ __forceinline__ void loss_and_kl_coef(int mode, const float* S, float n,
                                                 const smx_ppo_ctrl_t* __restrict__ ctrl,
                                                 float& loss, float& c_kl) {
    const float surr_mean = S[0] / n;
    const float kl_mean = S[2] / n;
    c_kl = 0.f;
    if (mode == SMX_PPO_CLIP) {
        loss = S[1] / n;
    } else {
        const float beta = ctrl->beta, eta = ctrl->eta;
        const double kt2 = 2.0 * (double)ctrl->kl_target;
        loss = surr_mean + beta * kl_mean;                              // ppo.py:272
        c_kl = beta;
        if ((double)kl_mean - kt2 > 0.0) {                              // ppo.py:275-276
            const float d = kl_mean - (float)kt2;
            loss += eta * (d * d);
            c_kl += 2.0f * eta * d;
        }
    }
}

// one thread: the epoch's statistics, the KL early exit and the step counters
__device__ __forceinline__ void write_policy_scalars(const float* S, float n, float loss, float c_kl,
                                                     const float* __restrict__ log_var, int A,
                                                     smx_ppo_ctrl_t* __restrict__ ctrl,
                                                     int check_stop, int will_update,
                                                     float* __restrict__ dlogvar_sumsq,
                                                     float* __restrict__ stats) {
    const float inv_n = 1.0f / n;
    const float kl_mean = S[2] / n;
    float ls = 0.f, dq = 0.f;
    for (int a = 0; a < A; ++a) {
        ls += logf(expf(log_var[a]));
        const float g = (S[8 + a] + c_kl * S[8 + A + a]) * inv_n;
        dq += g * g;
    }
    if (dlogvar_sumsq) *dlogvar_sumsq = dq;
    stats[SMX_PS_SURR] = S[0] / n;
    stats[SMX_PS_LOSS] = loss;
    // ppo_net.py:72 (sic): 0.5 * sum(log std) + 0.5 * log(2 pi e) * d
    stats[SMX_PS_ENTROPY] = 0.5f * ls + (float)(0.5 * 2.8378770664093453 * (double)A);
    stats[SMX_PS_KL] = kl_mean;
    stats[SMX_PS_LB] = S[3] / n;
    stats[SMX_PS_ISW] = S[4] / n;
    stats[SMX_PS_REFBEH] = S[5] / n;
    int stop = 0;
    if (check_stop && (double)kl_mean > 4.0 * (double)ctrl->kl_target) stop = 1;  // ppo.py:556
    if (stop) {
        ctrl->stop_flag = 1;
    } else if (will_update) {
        ctrl->adam_step_actor += 1;
        ctrl->epochs_done += 1;
    }
}

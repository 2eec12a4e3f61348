// Fused PPO minibatch: gather -> MLP forward -> PPO-clip loss -> MLP backward (data AND weight gradients) in ONE launch.
// Replaces memory.sample (xuance/common/memory_tools.py:267-287) and the forward / loss / backward of
// PPO_Learner.update (xuance/torch/learners/policy_gradient/ppo_learner.py:46-62) for an actor-critic MLP with a
// categorical head; xrl_grad_reduce + xrl_adam_step complete the optimiser step.
//
// A workgroup (8 waves) owns a 32-row tile of the minibatch.  Every activation level, every gradient level, the
// first-layer / head parameters and the biases live in LDS ("LDS-staged minibatch tiles"); only the big middle-layer
// weights stream from global memory (forward: W[N][K]; backward-data: the transposed copy W^T[K][N], so both use the
// same k-contiguous B-fragment path of mlp_tile.h).  Weight gradients of a middle layer are 32x32 MFMA tiles whose
// reduction dimension is the tile's 32 rows: wave w accumulates dW rows [32w, 32w+32) x 128 columns in 4 accumulators
// (64 VGPRs) and writes them straight into this workgroup's gradient slab, i.e. one deterministic partial per workgroup;
// the small gradients (first layer, heads, biases) are VALU reductions over the 32 rows.
// Per 32-row tile of the CartPole net (4-128-{128-2,128-1}): forward 2.2 MFLOP, backward 4.3 MFLOP on the matrix
// cores; traffic per workgroup ~0.26 MB of weights in, 0.136 MB of gradient slab out.
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"

namespace xrl {

// dW[N][K] (+)= dZ[32][N]^T . H[32][K] for one middle layer; dZ / H are LDS tiles, the result goes to the slab.
// wave w owns output row tiles w, w+8, ...; column tiles are processed 4 at a time (4 accumulators).
__device__ __forceinline__ void tile_weight_grad(const float* dz, int ld_dz, const float* hin, int ld_h, int N, int K,
                                                 float* __restrict__ dW /* [N][K] in the slab */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = (N + 31) / 32, k_tiles = (K + 31) / 32;
    for (int nt = wave; nt < n_tiles; nt += NW) {
        const int n0 = nt * 32;
        for (int kt0 = 0; kt0 < k_tiles; kt0 += 4) {
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            const float* arow = dz + lh * ld_dz + min(n0 + li, N - 1);         // A[i = n][k = row]: dZ[row][n]
#pragma unroll 4
            for (int s = 0; s < FT / 2; ++s) {                                  // rows 2s + lh
                const float a = arow[2 * s * ld_dz];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kc = (kt0 + t) * 32 + li;
                    const float b = (kt0 + t < k_tiles) ? hin[(2 * s + lh) * ld_h + min(kc, K - 1)] : 0.f;   // B[k = row][j]
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = (kt0 + t) * 32 + li;
                if (kt0 + t < k_tiles && col < K) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < N) dW[(size_t)row * K + col] = acc[t][r];
                    }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(FUSED_THREADS) ppo_fused_kernel(xrl_ppo_fused_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double scratch[16];
    __shared__ float s_row[FT][8];         // per-row scalars: act, ret, adv, old_logp
    constexpr int D = 4;
    const int tid = threadIdx.x, M = p.M, A = p.A;
    const int lane = tid & 63, wave = tid >> 6, li_ = lane & 31, lh_ = lane >> 5;
    const int m0 = blockIdx.x * FT;
    const int nL = p.n_layers, nH = p.n_head_layers, nLv = p.n_levels, end_mid = nL - nH;
    float* slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
    long long* dbg = p.dbg;
    int dbi = 0;
#define PSTAMP() do { if (dbg && tid == 0 && blockIdx.x == gridDim.x - 1) { dbg[dbi++] = clock64(); dbg[15] = dbi; } } while (0)
    PSTAMP();

    // ---- LDS carve: activation levels 1.., gradient levels 1.., split-K scratch, parameter cache (same image as the rollout)
    int lvl_off[XRL_FUSED_MAX_LEVELS], g_off[XRL_FUSED_MAX_LEVELS], lvl_ld[XRL_FUSED_MAX_LEVELS];
    int off = 0;
#pragma unroll
    for (int l = 0; l < XRL_FUSED_MAX_LEVELS; ++l) {
        lvl_ld[l] = l < nLv ? level_ld(p.level_width[l]) : 0;
        lvl_off[l] = off;
        if (l >= 1 && l < nLv) off += FT * lvl_ld[l];
    }
#pragma unroll
    for (int l = 0; l < XRL_FUSED_MAX_LEVELS; ++l) {
        g_off[l] = off;
        if (l >= 1 && l < nLv) off += FT * lvl_ld[l];
    }
    const int acts_end = off;
    float* red = lds + off;             off += NW * 32 * 33;
    off = (off + 3) / 4 * 4;
    const xrl_fused_layer_t& L0 = p.layers[0];
    const int c_w0 = off;               off += L0.N * 4;
    const int c_b0 = off;               off += (L0.N + 3) / 4 * 4;
    int c_bm[XRL_FUSED_MAX_LAYERS], c_wm[XRL_FUSED_MAX_LAYERS];
#pragma unroll
    for (int l = 0; l < XRL_FUSED_MAX_LAYERS; ++l) {
        c_bm[l] = off; c_wm[l] = -1;
        if (l >= 1 && l < end_mid) {
            off += (p.layers[l].N + 3) / 4 * 4;
            if (layer_small(p.layers[l].N, p.layers[l].K)) { c_wm[l] = off; off += p.layers[l].N * level_ld(p.layers[l].K); }
        }
    }
    const int KH = p.level_width[nLv - 2], NH = p.level_width[nLv - 1], ldH = level_ld(KH);
    const int c_wh = off;               off += NH * ldH;
    const int c_bh = off;               off += (NH + 3) / 4 * 4;
    const int pc_base = c_w0, pc_floats = off - c_w0;

    // ---- issue every load now: first big layer's B-fragments, this thread's row (gather), the parameter-cache image
    float4 pf[PD];
    int pf_layer = -1;
#pragma unroll
    for (int l = 1; l < XRL_FUSED_MAX_LAYERS; ++l) {
        if (pf_layer < 0 && l < end_mid && c_wm[l] < 0) {
            const xrl_fused_layer_t& L = p.layers[l];
            const float* Wg = p.params + L.w_off;
            if ((L.N + 31) / 32 >= NW && (L.K & 7) == 0 && ((reinterpret_cast<uintptr_t>(Wg) & 15) == 0)) {
                pf_layer = l;
                const int kq = L.K / 8;
                const float* wrow = Wg + (size_t)min(wave * 32 + li_, L.N - 1) * L.K + 4 * lh_;
#pragma unroll
                for (int q = 0; q < PD; ++q) pf[q] = q < kq ? *reinterpret_cast<const float4*>(wrow + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int r = tid >> 4, sub = tid & 15, m_row = m0 + r;
    const bool row_ok = m_row < M;
    size_t src = 0;
    if (row_ok) {                                                       // (env, t) = divmod(idx, T); field[t][env]
        const int64_t fl = p.idx[m_row];
        const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
        src = (size_t)t * p.n_envs + env;
    }
    float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok) xrow = *reinterpret_cast<const float4*>(p.f_obs + src * D);
    float g_act = 0.f, g_ret = 0.f, g_adv = 0.f, g_lp = 0.f;
    if (row_ok && sub == 0) { g_act = p.f_act[src]; g_ret = p.f_ret[src]; g_adv = p.f_adv[src]; g_lp = p.f_logp[src]; }
    float4 img[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i4 = (tid + j * FUSED_THREADS) * 4;
        img[j] = i4 < pc_floats ? *reinterpret_cast<const float4*>(p.cache_image + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i4 = (tid + 4 * FUSED_THREADS) * 4; i4 < pc_floats; i4 += FUSED_THREADS * 4)
        *reinterpret_cast<float4*>(&lds[pc_base + i4]) = *reinterpret_cast<const float4*>(p.cache_image + i4);
    for (int i = tid; i < acts_end; i += FUSED_THREADS) lds[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i4 = (tid + j * FUSED_THREADS) * 4;
        if (i4 < pc_floats) *reinterpret_cast<float4*>(&lds[pc_base + i4]) = img[j];
    }
    if (sub == 0) {
        float adv = g_adv;
        if (p.stats && row_ok) adv = __fdiv_rn(__fsub_rn(adv, p.stats[0]), p.stats[1] + 1e-8f);     // memory_tools.py:281-282
        s_row[r][0] = g_act; s_row[r][1] = g_ret; s_row[r][2] = adv; s_row[r][3] = g_lp;
    }
    __syncthreads();
    PSTAMP();

    // ================================================================== forward
    {   // first layer on the VALU (K = 4): same k-ordered fma chain as the MFMA path
        float* o1 = lds + lvl_off[L0.out_level] + L0.out_off + r * lvl_ld[L0.out_level];
        XRL_ACT_DISPATCH(L0.act,
            for (int c = sub; c < L0.N; c += 16) {
                const float4 w = *reinterpret_cast<const float4*>(&lds[c_w0 + c * 4]);
                float acc = __fmaf_rn(xrow.x, w.x, 0.f);
                acc = __fmaf_rn(xrow.y, w.y, acc);
                acc = __fmaf_rn(xrow.z, w.z, acc);
                acc = __fmaf_rn(xrow.w, w.w, acc);
                o1[c] = act_apply_c<ACT>(acc + lds[c_b0 + c]);
            })
    }
    __syncthreads();
#pragma unroll
    for (int l = 1; l < XRL_FUSED_MAX_LAYERS; ++l) {
        if (l < end_mid) {
            const xrl_fused_layer_t& L = p.layers[l];
            fused_layer(p.params + L.w_off, c_wm[l] >= 0 ? lds + c_wm[l] : nullptr, level_ld(L.K), lds + c_bm[l], L.K, L.N, L.act,
                        lds + lvl_off[L.in_level] + L.in_off, lvl_ld[L.in_level],
                        lds + lvl_off[L.out_level] + L.out_off, lvl_ld[L.out_level], red, pf, l == pf_layer);
        }
    }
    if (NH <= 8)
        narrow_layer_valu(lds + c_wh, ldH, lds + c_bh, KH, NH, XRL_ACT_NONE, lds + lvl_off[nLv - 2], lvl_ld[nLv - 2],
                          lds + lvl_off[nLv - 1], lvl_ld[nLv - 1]);
    else
        fused_layer(nullptr, lds + c_wh, ldH, lds + c_bh, KH, NH, XRL_ACT_NONE, lds + lvl_off[nLv - 2], lvl_ld[nLv - 2],
                    lds + lvl_off[nLv - 1], lvl_ld[nLv - 1], red, pf, false);

    PSTAMP();
    // ================================================================== loss (one thread per row)
    const int ldh = lvl_ld[nLv - 1];
    float* heads = lds + lvl_off[nLv - 1];
    float* dheads = lds + g_off[nLv - 1];
    double acc_s = 0.0, acc_c = 0.0, acc_e = 0.0, acc_v = 0.0, acc_n = 0.0;
    if (tid < FT) {
        const int m = m0 + tid;
        float* dq = dheads + tid * ldh;
        if (m < M) {
            const float* o = heads + tid * ldh;
            const float invM = 1.f / (float)M;
            const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
            const int a = (int)s_row[tid][0];
            const float ret = s_row[tid][1], adv = s_row[tid][2], oldlp = s_row[tid][3], v = o[A];
            float mx = o[0];
            for (int j = 1; j < A; ++j) mx = fmaxf(mx, o[j]);
            float se = 0.f;
            for (int j = 0; j < A; ++j) se += expf(o[j] - mx);
            const float lse = mx + logf(se);
            const float logp = o[a] - lse;
            float ent = 0.f;
            for (int j = 0; j < A; ++j) { const float l = o[j] - lse; ent -= expf(l) * l; }
            const Surrogate s = surrogate(logp, oldlp, adv, lo, hi, invM);
            const float ce = p.ent_coef * invM;
            for (int j = 0; j < A; ++j) {
                const float l = o[j] - lse, pj = expf(l);
                dq[j] = s.dlogp * ((j == a ? 1.f : 0.f) - pj) + ce * pj * (l + ent);
            }
            const float dv = v - ret;
            dq[A] = p.vf_coef * 2.f * dv * invM;
            acc_s = (double)fminf(s.s1, s.s2); acc_n = s.clipped; acc_c = (double)dv * dv; acc_e = ent; acc_v = v;
            if (p.diag) { p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2; }
        } else {
            for (int j = 0; j <= A; ++j) dq[j] = 0.f;                  // rows beyond the minibatch contribute nothing
        }
    }
    if (wave == 0) {
        acc_s = wave_sum(acc_s); acc_c = wave_sum(acc_c); acc_e = wave_sum(acc_e); acc_v = wave_sum(acc_v); acc_n = wave_sum(acc_n);
        if (lane == 0) {
            double* q = p.partials + (size_t)blockIdx.x * 8;
            q[0] = acc_s; q[1] = acc_c; q[2] = acc_e; q[3] = acc_v; q[4] = acc_n; q[5] = 0; q[6] = 0; q[7] = 0;
        }
    }
    __syncthreads();
    PSTAMP();

    // ================================================================== backward
    // ---- heads: dW_h[j][k] = sum_rows dZh[row][j] * H[row][k] (only the blocks that belong to a head), db_h, and
    //      dH = dZh . W_h, times act'(H) of the layer that produced H (all layers writing one level share an activation)
    {
        const float* hprev = lds + lvl_off[nLv - 2];
        const int ldp = lvl_ld[nLv - 2];
        float* dprev = lds + g_off[nLv - 2];
        int pact = L0.act;
#pragma unroll
        for (int l = 1; l < XRL_FUSED_MAX_LAYERS; ++l)
            if (l < end_mid && p.layers[l].out_level == nLv - 2) pact = p.layers[l].act;
#pragma unroll
        for (int l = 1; l < XRL_FUSED_MAX_LAYERS; ++l) {
            if (l >= end_mid && l < nL) {
                const xrl_fused_layer_t& L = p.layers[l];
                for (int j = 0; j < L.N; ++j) {
                    for (int k = tid; k < L.K; k += FUSED_THREADS) {
                        float acc = 0.f;
#pragma unroll 8
                        for (int rr = 0; rr < FT; ++rr) acc += dheads[rr * ldh + L.out_off + j] * hprev[rr * ldp + L.in_off + k];
                        slab[L.w_off + j * L.K + k] = acc;
                    }
                }
                if (tid < L.N) {                                   // (bias gradients: the tile's rows added in double, rounded once -- as
                    double acc = 0.0;                              //  ppo_trunk_kernel does since round 6: the two kernels carry the same bits)
                    for (int rr = 0; rr < FT; ++rr) acc += (double)dheads[rr * ldh + L.out_off + tid];
                    slab[L.b_off + tid] = (float)acc;
                }
            }
        }
        // thread -> (row group, column): column k = tid mod KH' with KH' = 512 / rows-in-flight, no divisions
        const int cols_per_pass = KH < FUSED_THREADS ? KH : FUSED_THREADS;
        const int rows_per_pass = FUSED_THREADS / cols_per_pass;       // KH = 256 -> 2 rows at a time
        const int rsub = tid / cols_per_pass, k0c = tid - rsub * cols_per_pass;
        XRL_ACT_DISPATCH(pact,
            for (int k = k0c; k < KH; k += cols_per_pass) {
                float w[8];
                _Pragma("unroll") for (int j = 0; j < 8; ++j) w[j] = j < NH ? lds[c_wh + j * ldH + k] : 0.f;
                for (int rr = rsub; rr < FT; rr += rows_per_pass) {
                    float acc = 0.f;
                    _Pragma("unroll") for (int j = 0; j < 8; ++j) if (j < NH) acc += dheads[rr * ldh + j] * w[j];
                    for (int j = 8; j < NH; ++j) acc += dheads[rr * ldh + j] * lds[c_wh + j * ldH + k];
                    dprev[rr * ldp + k] = acc * act_grad_c<ACT>(hprev[rr * ldp + k]);
                }
            })
    }
    __syncthreads();
    PSTAMP();
    // ---- middle layers, last to first: dW (MFMA, register accumulators -> slab), db (column sums), dH (MFMA on W^T)
#pragma unroll
    for (int l = XRL_FUSED_MAX_LAYERS - 1; l >= 1; --l) {
        if (l < end_mid) {
            const xrl_fused_layer_t& L = p.layers[l];
            const float* dz = lds + g_off[L.out_level] + L.out_off;
            const int ldz = lvl_ld[L.out_level];
            const float* hin = lds + lvl_off[L.in_level] + L.in_off;
            const int ldi = lvl_ld[L.in_level];
            tile_weight_grad(dz, ldz, hin, ldi, L.N, L.K, slab + L.w_off);
            PSTAMP();
            for (int j = tid; j < L.N; j += FUSED_THREADS) {
                double acc = 0.0;
#pragma unroll 8
                for (int rr = 0; rr < FT; ++rr) acc += (double)dz[rr * ldz + j];
                slab[L.b_off + j] = (float)acc;
            }
            // dH_in = dZ . W  ==  "forward" with the transposed weights W^T[K][N]; epilogue multiplies by act'(H_in)
            int pact = L0.act;
#pragma unroll
            for (int l2 = 1; l2 < XRL_FUSED_MAX_LAYERS; ++l2)
                if (l2 < l && p.layers[l2].out_level == L.in_level) pact = p.layers[l2].act;
            fused_layer(p.params_t + L.w_off, nullptr, 0, nullptr, L.N, L.K, pact, dz, ldz,
                        lds + g_off[L.in_level] + L.in_off, ldi, red, pf, false, hin, ldi);
        }
    }
    PSTAMP();
    // ---- first layer: dW0[c][k] = sum_rows dZ1[row][c] * x[row][k], db0 (VALU; x rows are re-read from LDS scratch)
    {
        float* xs = red;                                                // [32][4] scratch
        if (sub == 0) *reinterpret_cast<float4*>(&xs[r * 4]) = xrow;
        __syncthreads();
        const float* dz1 = lds + g_off[L0.out_level] + L0.out_off;
        const int ld1 = lvl_ld[L0.out_level];
        for (int i = tid; i < L0.N * 4; i += FUSED_THREADS) {
            const int c = i >> 2, k = i & 3;
            float acc = 0.f;
            for (int rr = 0; rr < FT; ++rr) acc += dz1[rr * ld1 + c] * xs[rr * 4 + k];
            slab[L0.w_off + i] = acc;
        }
        for (int c = tid; c < L0.N; c += FUSED_THREADS) {
            float acc = 0.f;
            for (int rr = 0; rr < FT; ++rr) acc += dz1[rr * ld1 + c];
            slab[L0.b_off + c] = acc;
        }
    }
    PSTAMP();
}

// params_t[w_off + k*N + n] = params[w_off + n*K + k] for every middle layer
__global__ void __launch_bounds__(256) transpose_mid_kernel(xrl_ppo_fused_t p, float* __restrict__ out) {
    const int end_mid = p.n_layers - p.n_head_layers;
    for (int l = 1; l < end_mid; ++l) {
        const xrl_fused_layer_t& L = p.layers[l];
        const int total = L.N * L.K;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int k = i / L.N, n = i - k * L.N;                     // coalesced writes
            out[L.w_off + i] = p.params[L.w_off + (size_t)n * L.K + k];
        }
    }
}

// fragment-ordered copies of the first middle layer (see include/xrl_hip.h)
__global__ void __launch_bounds__(256) pack_mid_frags_kernel(xrl_ppo_fused_t p, float* __restrict__ frag) {
    const xrl_fused_layer_t& L = p.layers[1];
    const int N = L.N, K = L.K, kq = K / 8, nq = N / 8;
    const float* W = p.params + L.w_off;
    const int total = N * K;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int s = i & 3, l = (i >> 2) & 63;
        {   // forward section: chunk q of tile t lives in slot frag_slot(q, t, kq, 1)
            const int q = (i >> 8) % kq, t = (i >> 8) / kq;
            frag[((size_t)(t * kq + frag_slot(q, t, kq, 1)) * 64 + l) * 4 + s] = W[(size_t)(t * 32 + (l & 31)) * K + q * 8 + 4 * (l >> 5) + s];
        }
        {   // backward section: slot frag_slot(q, t, nq, 2) (two waves share a tile, taking even / odd chunks)
            const int q = (i >> 8) % nq, t = (i >> 8) / nq;
            frag[total + ((size_t)(t * nq + frag_slot(q, t, nq, 2)) * 64 + l) * 4 + s] = W[(size_t)(q * 8 + 4 * (l >> 5) + s) * K + t * 32 + (l & 31)];
        }
    }
}

__global__ void __launch_bounds__(256) pack_transitions_kernel(const float4* __restrict__ obs, const float* __restrict__ act,
                                                               const float* __restrict__ ret, const float* __restrict__ adv,
                                                               const float* __restrict__ logp, float4* __restrict__ out,
                                                               int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        out[2 * i] = obs[i];
        out[2 * i + 1] = make_float4(act[i], ret[i], adv[i], logp[i]);
    }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float4* __restrict__ packed, const int64_t* __restrict__ idx,
                                                          float4* __restrict__ out, int64_t count, int n_envs, int T) {
    // two threads per record (one float4 each)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fl = idx[i >> 1];
        const int env = (int)(fl / T), t = (int)(fl - (int64_t)env * T);
        out[i] = packed[2 * ((size_t)t * n_envs + env) + (i & 1)];
    }
}

static size_t ppo_fused_lds_bytes(const xrl_ppo_fused_t& p) {
    size_t floats = 0;
    for (int l = 1; l < p.n_levels; ++l) floats += 2 * (size_t)FT * level_ld(p.level_width[l]);
    floats += NW * 32 * 33 + 4;
    floats += (size_t)p.layers[0].N * 4 + (p.layers[0].N + 3) / 4 * 4;
    for (int l = 1; l < p.n_layers - p.n_head_layers; ++l) {
        floats += (p.layers[l].N + 3) / 4 * 4;
        if (layer_small(p.layers[l].N, p.layers[l].K)) floats += (size_t)p.layers[l].N * level_ld(p.layers[l].K);
    }
    const int NH = p.level_width[p.n_levels - 1], KH = p.level_width[p.n_levels - 2];
    floats += (size_t)NH * level_ld(KH) + (NH + 3) / 4 * 4;
    return floats * sizeof(float);
}

}  // namespace xrl

namespace xrl {
bool g_fast_enabled_ppo = true;                      // xrl_set_fast_kernels (tests): the specialised kernel families on / off
bool ppo_trunk_eligible(const xrl_ppo_fused_t& p);
int launch_ppo_trunk(const xrl_ppo_fused_t& p, const xrl_opt_chain_t* o, hipStream_t stream);
int init_ppo_trunk();
int init_ppo_fused();
bool ppo_trunk_bx_eligible(const xrl_ppo_fused_t& p);           // csrc/ppo_trunk_bx.hip: the CartPole class on exact 3-way bf16 splits
int launch_ppo_trunk_bx(const xrl_ppo_fused_t& p, hipStream_t stream);
int init_ppo_trunk_bx();
}
using namespace xrl;

int xrl::init_ppo_fused() {
    if (int rc = init_ppo_trunk()) return rc;
    if (int rc = init_ppo_trunk_bx()) return rc;
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fused_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    return XRL_OK;
}

extern "C" int xrl_ppo_fused_minibatch(const xrl_ppo_fused_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_ppo_fused_t& p = *pp;
    XRL_CHECK_ARG(p.params != nullptr);
    XRL_CHECK_ARG(p.f_obs && p.f_act && p.f_ret && p.f_adv && p.f_logp && p.idx && p.slabs && p.partials);
    XRL_CHECK_ARG(p.M > 0 && p.n_envs > 0 && p.T > 0);
    // the shared-trunk family D-128-{128-A | 128-1} (D <= 24, A <= 8, categorical | Gaussian): (tile, role) workgroups, 32- or 64-row tiles
    if (p.l0_fold_off > 0) {
        if (!g_fast_enabled_ppo || !ppo_trunk_eligible(p)) { set_error("xrl_ppo_fused_minibatch: a fold region was given but the network is not of the shared-trunk family (csrc/ppo_trunk.hip)"); return XRL_EINVAL; }
        if (ppo_trunk_bx_eligible(p)) return launch_ppo_trunk_bx(p, as_stream(stream));
        return launch_ppo_trunk(p, nullptr, as_stream(stream));
    }
    XRL_CHECK_ARG(p.params_t && p.cache_image && (reinterpret_cast<uintptr_t>(p.cache_image) & 15) == 0);
    XRL_CHECK_ARG(p.D == 4 && p.A >= 2 && p.dist == 0);
    XRL_CHECK_ARG(p.n_layers >= 2 && p.n_layers <= XRL_FUSED_MAX_LAYERS && p.n_levels >= 3 && p.n_levels <= XRL_FUSED_MAX_LEVELS);
    XRL_CHECK_ARG(p.n_head_layers >= 1 && p.n_head_layers < p.n_layers && p.layers[0].K == 4 && p.layers[0].in_level == 0);
    XRL_CHECK_ARG(p.level_width[0] == 4 && p.level_width[p.n_levels - 1] == p.A + 1);
    for (int l = 1; l < p.n_layers - p.n_head_layers; ++l) XRL_CHECK_ARG(p.layers[l].N % 32 == 0 && p.layers[l].K % 32 == 0);
    const size_t lds_bytes = ppo_fused_lds_bytes(p);
    XRL_CHECK_ARG(lds_bytes <= 156 * 1024);
    const int n_tiles = (p.M + FT - 1) / FT;
    hipLaunchKernelGGL(ppo_fused_kernel, dim3(n_tiles), dim3(FUSED_THREADS), lds_bytes, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_pack_transitions(const float* f_obs, const float* f_act, const float* f_ret, const float* f_adv,
                                    const float* f_logp, float* packed, int64_t count, xrl_stream_t stream) {
    XRL_CHECK_ARG(f_obs && f_act && f_ret && f_adv && f_logp && packed && count > 0);
    XRL_CHECK_ARG(((reinterpret_cast<uintptr_t>(f_obs) | reinterpret_cast<uintptr_t>(packed)) & 15) == 0);
    int nb = (int)((count + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(pack_transitions_kernel, dim3(nb), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(f_obs), f_act, f_ret, f_adv, f_logp, reinterpret_cast<float4*>(packed), count);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_gather_rows(const float* packed, const int64_t* idx, float* out, int64_t count, int n_envs, int T,
                               xrl_stream_t stream) {
    XRL_CHECK_ARG(packed && idx && out && count > 0 && n_envs > 0 && T > 0);
    XRL_CHECK_ARG(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    int nb = (int)((2 * count + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(packed), idx,
                       reinterpret_cast<float4*>(out), count, n_envs, T);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_pack_mid_frags(const xrl_ppo_fused_t* pp, float* frag, int64_t frag_floats, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp && frag && pp->params && pp->n_layers - pp->n_head_layers >= 2);
    const xrl_fused_layer_t& L = pp->layers[1];
    XRL_CHECK_ARG(L.N % 32 == 0 && L.K % 32 == 0 && frag_floats >= 2 * (int64_t)L.N * L.K);
    hipLaunchKernelGGL(pack_mid_frags_kernel, dim3(64), dim3(256), 0, as_stream(stream), *pp, frag);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_transpose_mid(const xrl_ppo_fused_t* pp, float* params_t, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp && params_t && pp->params && pp->n_layers >= 2 && pp->n_layers <= XRL_FUSED_MAX_LAYERS);
    hipLaunchKernelGGL(transpose_mid_kernel, dim3(64), dim3(256), 0, as_stream(stream), *pp, params_t);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

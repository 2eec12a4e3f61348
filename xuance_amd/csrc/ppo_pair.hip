// Role-split fused PPO minibatch on 64-ROW tiles for the actor-critic 4-128-{128-2, 128-1}: the arithmetic of ppo_split_kernel
// (csrc/ppo_split.hip: one workgroup per (tile, branch); the roles write disjoint slab regions, the critic role's first-layer
// gradient goes to the fold region) with TWO 32-row MFMA blocks per workgroup.
//
// Why: per minibatch of 8 192 rows ppo_fast_kernel has 256 workgroups that each pull the whole stacked branch layer twice
// (128 KB forward fragments + 128 KB backward fragments at the ~10 B/clk a CU gets out of L2) and write a 136 KB gradient slab --
// 35 MB of slabs that the optimiser launch reads back (profiles/r02_i_ppo_c2_pmc_hbm.json: 38 MB written, 41 MB read per
// minibatch for ~0.5 MB of algorithmic bytes).  With a (64-row tile, role) decomposition the same 256 workgroups
//   * stream HALF the weights each (their role's 64 KB, forward and backward section), shared by the two MFMA blocks:
//     waves w and w + 4 own the same 32 output columns for rows [0, 32) and [32, 64) and hit the same lines in L1;
//   * keep all eight waves on the matrix cores in the forward and backward-data phases (ppo_split: four);
//   * accumulate the weight gradient over 64 rows in the MFMA accumulators: 128 slabs of 136 KB per minibatch instead of 256;
//   * run the latency chain of a tile (gather -> first layer -> branch layer -> heads / loss -> gradients) once per 64 rows.
// Same per-element arithmetic as ppo_fast / ppo_split (same MFMA k-order inside a 32-row block); sums over rows that used to be
// formed by the slab reduction (two 32-row partials) are now formed in the accumulators (rows 0..63 in order).
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward).
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"

namespace xrl {

typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
constexpr int PH = 128;                      // hidden width (trunk and each branch)
constexpr int PLD = PH + 4;                  // row stride of every LDS level
constexpr int PT = 64;                       // rows per workgroup
// packed image layout (pack_rollout_cache_kernel) for 4-128-256-{2|1} -- see ppo_fast.hip
constexpr int QI_W0_ = 0, QI_B0_ = 4 * PH, QI_BM_ = QI_B0_ + PH, QI_WH_ = QI_BM_ + 2 * PH, QI_LDH_ = 2 * PH + 4, QI_BH_ = QI_WH_ + 3 * QI_LDH_;
constexpr int QI_FLOATS_ = QI_BH_ + 4;
constexpr int PP_LDS_FLOATS = 4 * PT * PLD + 3 * PT * 4 + QI_FLOATS_;
constexpr int PP_LDS_BYTES = PP_LDS_FLOATS * 4 + PT * 5 * 8;           // 144 KB: one workgroup per CU

__device__ __forceinline__ float row8_sum(float v) {                   // sum over the 8 lanes of a row (lanes 8 r .. 8 r + 7)
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}

template <int ACT>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_pair_kernel(xrl_ppo_fused_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                   // [64][132] first hidden level
    float* h2 = h1 + PT * PLD;                         // [64][132] this role's branch level
    float* g2 = h2 + PT * PLD;                         // [64][132] dLoss/d(pre-activation of h2)
    float* xb = g2 + PT * PLD;                         // [64][132] g1 (this role's part)
    float* xs = xb + PT * PLD;                         // [64][4] gathered observations
    float* dzh = xs + PT * 4;                          // [64][4] dLoss/d(logits | value)
    float* rsc = dzh + PT * 4;                         // [64][4] gathered act | ret | adv | old_logp
    float* pimg = rsc + PT * 4;                        // [QI_FLOATS_] packed small-parameter image
    double* rowstat = reinterpret_cast<double*>(pimg + QI_FLOATS_);   // [5][64] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_fused_t)>();
    constexpr int D = 4;
    const int tid = threadIdx.x, M = p.M;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cblk = wave & 3, rblk = wave >> 2;       // this wave's 32-column block / 32-row block in the MFMA phases
    const int tile = blockIdx.x >> 1, role = blockIdx.x & 1;
    const bool actor = role == 0;
    const int cb = role * PH;                          // this role's first column of the stacked branch level
    const int m0 = tile * PT;
    const int r = tid >> 3, sub = tid & 7, m_row = m0 + r;     // VALU phases: 8 threads per row
    const bool row_ok = m_row < M;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const float* img = p.cache_image;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];

    // diagnostics (p.dbg != NULL; tools/probe_pair_phases.py): shader-clock stamps of the last workgroup's phases in dbg[0..11], and
    // the 100 MHz real-time counter at the start / end of EVERY workgroup in dbg[16 + 2 b], dbg[17 + 2 b] (launch skew, tail)
    long long* dbg = p.dbg;
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - 1;
#define PSTAMP(k) do { if (dbg_me) dbg[k] = clock64(); } while (0)
    if (dbg && tid == 0) dbg[16 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    PSTAMP(0);
    // ================= loads: gather (wave 7, lane = row), parameter image, this role's 64 KB of W1 B-fragments (every wave:
    //                   waves w and w + 4 fetch the same lines)
    float4 pf[PD];
    if (wave == 7) {
        const int m = m0 + lane;
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.f_rows) {
            if (m < M) {
                const float4* rec = reinterpret_cast<const float4*>(p.f_rows) + (size_t)m * 2;
                xr = rec[0]; sc = rec[1];
            }
        } else if (m < M) {
            const int64_t fl = p.idx[m];
            const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
            const size_t src = (size_t)t * p.n_envs + env;
            if (p.f_packed) {
                const float4* rec = reinterpret_cast<const float4*>(p.f_packed) + src * 2;
                xr = rec[0]; sc = rec[1];
            } else {
                xr = *reinterpret_cast<const float4*>(p.f_obs + src * D);
                sc = make_float4(p.f_act[src], p.f_ret[src], p.f_adv[src], p.f_logp[src]);
            }
        }
        *reinterpret_cast<float4*>(xs + lane * 4) = xr;
        *reinterpret_cast<float4*>(rsc + lane * 4) = sc;
    }
    float st_mean = 0.f, st_std = 1.f;
    if (p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    float4 imgv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < QI_FLOATS_ / 4) imgv = *reinterpret_cast<const float4*>(img + tid * 4);
    {
        const int t = 4 * role + cblk;                                   // tile of the stacked 256-row W1
        const float* base = p.frag_image + ((size_t)t * (PH / 8) * 64 + lane) * 4;
#pragma unroll
        for (int q = 0; q < PD; ++q) pf[q] = *reinterpret_cast<const float4*>(base + frag_slot(q, t, PH / 8, 1) * 256);
    }
    if (tid < QI_FLOATS_ / 4) *reinterpret_cast<float4*>(pimg + tid * 4) = imgv;
    lds_barrier();                                                                                   // #0 gathered rows
    PSTAMP(1);
    const float4 xrow = *reinterpret_cast<const float4*>(xs + r * 4);
    const float4 rowsc = *reinterpret_cast<const float4*>(rsc + r * 4);
    const float g_act = rowsc.x, g_ret = rowsc.y, g_adv = rowsc.z, g_lp = rowsc.w;

    // ================= forward: first layer on the VALU (k-ordered fma chain == the MFMA result): 16 columns per thread.  The
    //                   eight threads of a row walk their 16 columns in orders rotated by `sub`, so that one instruction's reads of
    //                   the weight rows (float4 at 16-byte stride 64 floats between the subs: ONE bank group without the rotation,
    //                   an 8-way conflict) fall on eight different bank groups
    {
        float* dst = h1 + r * PLD + sub * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int cc = (k + sub) & 15;
            const float4 w = *reinterpret_cast<const float4*>(pimg + QI_W0_ + (sub * 16 + cc) * 4);
            const float b = pimg[QI_B0_ + sub * 16 + cc];
            float acc = __fmaf_rn(xrow.x, w.x, 0.f);
            acc = __fmaf_rn(xrow.y, w.y, acc);
            acc = __fmaf_rn(xrow.z, w.z, acc);
            acc = __fmaf_rn(xrow.w, w.w, acc);
            dst[cc] = act_apply_c<ACT>(acc + b);
        }
    }
    lds_barrier();                                                                                   // #1 h1
    PSTAMP(2);
    // ---- this role's branch layer 128 -> 128 on the matrix cores: wave (cblk, rblk) owns columns [32 cblk, +32) of rows [32 rblk, +32)
    {
        const float* arow = h1 + (rblk * 32 + li) * PLD + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            float4 af[PD / 2];
#pragma unroll
            for (int q = 0; q < PD / 2; ++q) af[q] = *reinterpret_cast<const float4*>(arow + (hq * 8 + q) * 8);
#pragma unroll
            for (int q = 0; q < PD / 2; ++q) { MFMA4(af[q], pf[hq * 8 + q], acc) }
        }
        const int col = cblk * 32 + li;
        const float bm = pimg[QI_BM_ + cb + col];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * PLD + col] = act_apply_c<ACT>(acc[rr] + bm);
        }
        // forward fragments consumed: the same registers take the BACKWARD section of the fragment copy for dH1 below (output tile
        // kt = cblk, this role's 16 n-chunks q = 16 role + i; xrl_pack_mid_frags) -- it has the head / loss / weight-gradient
        // phases to arrive
        {
            const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_image), 0, 2 * 2 * PH * PH * 4, 0x00020000);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                const int q = 16 * role + i;
                const pu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (2 * PH * PH + (cblk * 32 + frag_slot(q, cblk, 32, 2)) * 256) * 4, 0);
                pf[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                                                                   // #2 h2
    PSTAMP(3);

    // ================= head forward (VALU, 8 threads per row), this role's loss terms, head backward -- in registers
    // k-chunks q = sub + 8 i, i = 0..3 (the 32 float4 chunks of this role's 128 columns)
    float4 a[4], wa[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const float4*>(h2 + r * PLD + 4 * (sub + 8 * i));
        // merged head rows: 0, 1 = logits over columns [0, 128), 2 = value over columns [128, 256)
        wa[0][i] = *reinterpret_cast<const float4*>(pimg + QI_WH_ + (actor ? 0 : 2) * QI_LDH_ + cb + 4 * (sub + 8 * i));
        wa[1][i] = actor ? *reinterpret_cast<const float4*>(pimg + QI_WH_ + 1 * QI_LDH_ + 4 * (sub + 8 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float hv0, hv1;
    {
        float c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c0 += a[i].x * wa[0][i].x + a[i].y * wa[0][i].y + a[i].z * wa[0][i].z + a[i].w * wa[0][i].w;
            c1 += a[i].x * wa[1][i].x + a[i].y * wa[1][i].y + a[i].z * wa[1][i].z + a[i].w * wa[1][i].w;
        }
        hv0 = row8_sum(c0) + pimg[QI_BH_ + (actor ? 0 : 2)];
        hv1 = row8_sum(c1) + pimg[QI_BH_ + 1];
    }
    float dz0 = 0.f, dz1 = 0.f;                          // actor: d/d(logit 0), d/d(logit 1); critic: d/d(value), 0
    {
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        if (row_ok) {
            const float invM = 1.f / (float)M;
            if (actor) {
                float adv = g_adv;
                asm volatile("" : "+v"(st_std));
                if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);           // memory_tools.py:281-282
                const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
                const int act = (int)g_act;
                const float o[2] = {hv0, hv1};
                float mx = o[0];
                mx = fmaxf(mx, o[1]);
                float se = 0.f;
                se += expf(o[0] - mx); se += expf(o[1] - mx);
                const float lse = mx + logf(se);
                const float logp = (act == 0 ? o[0] : o[1]) - lse;
                float ent = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) { const float l = o[j] - lse; ent -= expf(l) * l; }
                const Surrogate s = surrogate(logp, g_lp, adv, lo, hi, invM);
                const float ce = p.ent_coef * invM;
                float dq[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float l = o[j] - lse, pj = expf(l);
                    dq[j] = s.dlogp * ((j == act ? 1.f : 0.f) - pj) + ce * pj * (l + ent);
                }
                dz0 = dq[0]; dz1 = dq[1];
                t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                if (p.diag && sub == 0) {
                    const int m = m_row;
                    p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                }
            } else {
                const float v = hv0, dv = v - g_ret;
                dz0 = p.vf_coef * 2.f * dv * invM;
                t_c = (double)dv * dv; t_v = v;
            }
        }
        if (sub == 0) {
            dzh[r * 4 + 0] = dz0; dzh[r * 4 + 1] = dz1;
            rowstat[0 * PT + r] = t_s; rowstat[1 * PT + r] = t_c; rowstat[2 * PT + r] = t_e; rowstat[3 * PT + r] = t_v; rowstat[4 * PT + r] = t_n;
        }
    }
    // dH2 = dZh . W_h, times act'(h2): this thread's four k-chunks
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 g;
        if (actor) {
            g.x = (dz0 * wa[0][i].x + dz1 * wa[1][i].x) * act_grad_c<ACT>(a[i].x);
            g.y = (dz0 * wa[0][i].y + dz1 * wa[1][i].y) * act_grad_c<ACT>(a[i].y);
            g.z = (dz0 * wa[0][i].z + dz1 * wa[1][i].z) * act_grad_c<ACT>(a[i].z);
            g.w = (dz0 * wa[0][i].w + dz1 * wa[1][i].w) * act_grad_c<ACT>(a[i].w);
        } else {
            g.x = (dz0 * wa[0][i].x) * act_grad_c<ACT>(a[i].x);
            g.y = (dz0 * wa[0][i].y) * act_grad_c<ACT>(a[i].y);
            g.z = (dz0 * wa[0][i].z) * act_grad_c<ACT>(a[i].z);
            g.w = (dz0 * wa[0][i].w) * act_grad_c<ACT>(a[i].w);
        }
        *reinterpret_cast<float4*>(g2 + r * PLD + 4 * (sub + 8 * i)) = g;
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    PSTAMP(4);

    // ================= backward
    // ---- loss terms of this (tile, role): one lane per row, wave sum; the actor fills surrogate / entropy / clip count, the critic
    //      the value terms -- the partials reduction adds the rows of all workgroups
    if (wave == 7) {
        double acc_s = rowstat[lane], acc_c = rowstat[PT + lane], acc_e = rowstat[2 * PT + lane], acc_v = rowstat[3 * PT + lane],
               acc_n = rowstat[4 * PT + lane];
        acc_s = wave_sum(acc_s); acc_c = wave_sum(acc_c); acc_e = wave_sum(acc_e); acc_v = wave_sum(acc_v); acc_n = wave_sum(acc_n);
        if (lane == 0) {
            double* q = p.partials + (size_t)blockIdx.x * 8;
            q[0] = acc_s; q[1] = acc_c; q[2] = acc_e; q[3] = acc_v; q[4] = acc_n; q[5] = 0; q[6] = 0; q[7] = 0;
        }
    }
    // ---- head weight / bias gradients and this role's branch-layer bias gradient: VALU reductions over the 64 rows
    if (tid < (actor ? 2 * PH : PH)) {
        const int j = tid >> 7, k = tid & (PH - 1);
        const float* hp = h2 + k;
        float acc = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < PT; ++rr) acc += dzh[rr * 4 + j] * hp[rr * PLD];
        slab[(actor ? La.w_off + j * PH : Lc.w_off) + k] = acc;
    } else if (tid >= 4 * 64 && tid < 4 * 64 + PH) {
        const int t = tid - 4 * 64;
        float acc0 = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < PT; ++rr) acc0 += g2[rr * PLD + t];
        slab[L1.b_off + cb + t] = acc0;
        if (t < (actor ? 2 : 1)) {
            float acc = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < PT; ++rr) acc += dzh[rr * 4 + t];
            slab[actor ? La.b_off + t : Lc.b_off] = acc;
        }
    }
    PSTAMP(5);
    // ---- dW1[n][k] = sum over the 64 rows of g2[row][n] * h1[row][k] for this role's 128 rows n: 4 x 4 tiles of 32 x 32, wave w
    //      owns n-tile (w & 3) and the k-tiles 2 (w >> 2), 2 (w >> 2) + 1; 32 chained MFMAs per tile, rows in order
    {
        const int nt = wave & 3, kt0 = 2 * (wave >> 2);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* arow = g2 + lh * PLD + nt * 32 + li;               // A[i = n][k = row]
        const float* brow = h1 + lh * PLD + kt0 * 32 + li;              // B[k = row][j]
#pragma unroll 8
        for (int s = 0; s < PT / 2; ++s) {
            const float av = arow[2 * s * PLD];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bv = brow[2 * s * PLD + t * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
        float* dW = slab + L1.w_off + (size_t)(cb + nt * 32) * PH;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)row * PH + (kt0 + t) * 32 + li] = acc[t][rr];
            }
    }
    PSTAMP(6);
    // ---- this role's part of dH1 = g2 . W1 (sum over its 128 rows n): wave (cblk, rblk) owns output columns [32 cblk, +32) of
    //      rows [32 rblk, +32); B operand = the backward fragments requested after the forward layer
    {
        const int k_out = cblk * 32 + li;
        const float* arow = g2 + (rblk * 32 + li) * PLD + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            float4 af[PD / 2];
#pragma unroll
            for (int i = 0; i < PD / 2; ++i) af[i] = *reinterpret_cast<const float4*>(arow + (hq * 8 + i) * 8);
#pragma unroll
            for (int i = 0; i < PD / 2; ++i) { MFMA4(af[i], pf[hq * 8 + i], acc) }
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            xb[row * PLD + k_out] = acc[rr] * act_grad_c<ACT>(h1[row * PLD + k_out]);
        }
    }
    lds_barrier();                                                                                   // #5 g1 (this role's part)
    PSTAMP(7);
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c] -- the actor's part into the slab's first-layer
    //      region, the critic's into the fold region behind the parameters (the reduction adds it onto the same columns)
    //      thread = (column c of g1, row half, pair of input components): 32 rows x 2 components each, halves met by a lane shuffle
    {
        float* dst = actor ? slab : slab + p.l0_fold_off;
        const int w_at = actor ? L0.w_off : 0, b_at = actor ? L0.b_off : 4 * PH;
        const int c = cblk * 32 + li, kp = 2 * rblk;                   // input components kp, kp + 1
        const float* gcol = xb + (lh * 32) * PLD + c;
        const float* xin = xs + (lh * 32) * 4 + kp;
        float a0 = 0.f, a1 = 0.f, ab = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < PT / 2; ++rr) {
            const float g = gcol[rr * PLD];
            const float2 x = *reinterpret_cast<const float2*>(xin + rr * 4);
            a0 += g * x.x; a1 += g * x.y; ab += g;
        }
        a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); ab += __shfl_xor(ab, 32, 64);
        if (lh == 0) {
            *reinterpret_cast<float2*>(dst + w_at + c * 4 + kp) = make_float2(a0, a1);
            if (rblk == 0) dst[b_at + c] = ab;
        }
    }
    PSTAMP(8);
    if (dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's stores are out
        dbg[17 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef PSTAMP
}

bool ppo_split_eligible(const xrl_ppo_fused_t& p);

// the 64-row form needs what the role-split kernel needs; the caller asks for it with tile_rows == 64 (its slab / partials layout
// differs: one slab per 64 rows)
bool ppo_pair_eligible(const xrl_ppo_fused_t& p) { return p.pad0 == 64 && ppo_split_eligible(p); }

int launch_ppo_pair(const xrl_ppo_fused_t& p, hipStream_t stream) {
    const int n_tiles = (p.M + PT - 1) / PT;
    XRL_ACT_DISPATCH(p.layers[0].act,
        hipLaunchKernelGGL(ppo_pair_kernel<ACT>, dim3(2 * n_tiles), dim3(FUSED_THREADS), PP_LDS_BYTES, stream, p);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

int init_ppo_pair() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_pair_kernel<XRL_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_pair_kernel<XRL_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_pair_kernel<XRL_ACT_LEAKY_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_pair_kernel<XRL_ACT_TANH>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_pair_kernel<XRL_ACT_SIGMOID>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
    return XRL_OK;
}

}  // namespace xrl

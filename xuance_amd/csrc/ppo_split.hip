// Role-split fused PPO minibatch for the actor-critic 4-128-{128-2, 128-1}: the work of ppo_fast_kernel for one 32-row tile,
// divided between TWO workgroups -- role 0 owns the actor branch (W1 rows 0..127, the two logit rows), role 1 the critic
// branch (W1 rows 128..255, the value row) -- that need nothing from each other:
//   * the shared first layer (4 -> 128) is recomputed by both (a few hundred fmas);
//   * every parameter of the branch layer and of the heads belongs to exactly one role, so the two roles of a tile write
//     DISJOINT regions of the same gradient slab;
//   * the gradient of the shared first layer is linear in dLoss/dh1 = (actor part) + (critic part): each role forms the
//     first-layer gradient of ITS part; the actor's goes to the slab's first-layer region, the critic's to a small fold region
//     behind the parameters (slab columns [l0_fold_off, l0_fold_off + 640)), which the reduction adds onto columns [0, 640).
// What it is for: SMALL minibatches.  At the headline size (8 192 rows = 256 tiles) ppo_fast_kernel already has one
// workgroup per CU; this kernel then runs two per CU (77 KB of LDS each) and -- measured with phase stamps,
// tools/probe_split_phases.py -- wins nothing: the time of such a kernel is the LATENCY of one workgroup's chain of phases
// (stream -> first layer -> 64 chained MFMAs -> heads / loss -> small gradients -> dW -> dH from the second fragment stream ->
// first-layer gradients), not matrix-pipe throughput, and halving a workgroup's work halves only some of those chains
// (in-loop 33.1 vs 32.4 us per minibatch, plus 2.6 us for the fold in the reduction).  With 16 envs (512-row minibatches,
// 16 tiles) the same split puts the minibatch on 32 CUs instead of 16 and shortens the dW / head / small-gradient phases:
// 17.8 vs 21.6 us per launch, 29.8 vs 34.1 us per minibatch with the optimiser launch.  PPO_Learner selects it for
// minibatches of at most 32 tiles.  Same arithmetic per element as ppo_fast_kernel (same MFMA k-order, same reduction
// trees); results differ from it only where a sum that used to be formed in LDS (dLoss/dh1 of both branches, the five loss
// terms of a tile) is now formed by the slab / partials reduction.
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward).
#include <cstdlib>
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"

namespace xrl {

typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
constexpr int SH = 128;                      // hidden width (trunk and each branch)
constexpr int SLD = SH + 4;                  // row stride of every LDS level
// packed image layout (pack_rollout_cache_kernel) for 4-128-256-{2|1} -- see ppo_fast.hip
constexpr int SI_W0 = 0, SI_B0 = 4 * SH, SI_BM = SI_B0 + SH, SI_WH = SI_BM + 2 * SH, SI_LDH = 2 * SH + 4, SI_BH = SI_WH + 3 * SI_LDH;
constexpr int SI_FLOATS = SI_BH + 4;
constexpr int SP_LDS_FLOATS = 4 * FT * SLD + 3 * FT * 4 + SI_FLOATS;
constexpr int SP_LDS_BYTES = SP_LDS_FLOATS * 4 + FT * 5 * 8;           // 77.1 KB: two workgroups per CU
constexpr int SP_FOLD = 4 * SH + SH;                                   // first-layer weights + bias

template <int CTRL>
__device__ __forceinline__ float sdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float srow16_sum(float v) {                 // == the xor butterfly 8,4,2,1 (rollout_fast.hip)
    v += sdpp<0x128>(v); v += sdpp<0x124>(v); v += sdpp<0x122>(v); v += sdpp<0x121>(v);
    return v;
}

template <int ACT>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_split_kernel(xrl_ppo_fused_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                   // [32][132] first hidden level
    float* h2 = h1 + FT * SLD;                         // [32][132] this role's branch level
    float* g2 = h2 + FT * SLD;                         // [32][132] dLoss/d(pre-activation of h2)
    float* xb = g2 + FT * SLD;                         // [32][132] g1 (this role's part)
    float* xs = xb + FT * SLD;                         // [32][4] gathered observations
    float* dzh = xs + FT * 4;                          // [32][4] dLoss/d(logits | value)
    float* rsc = dzh + FT * 4;                         // [32][4] gathered act | ret | adv | old_logp
    float* pimg = rsc + FT * 4;                        // [SI_FLOATS] packed small-parameter image
    double* rowstat = reinterpret_cast<double*>(pimg + SI_FLOATS);   // [5][32] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_fused_t)>();
    constexpr int D = 4;
    const int tid = threadIdx.x, M = p.M;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x >> 1, role = blockIdx.x & 1;
    if (p.pad0 && role != p.pad0 - 1) return;          // diagnostics (XRL_SPLIT_ONLY_ROLE): one role alone on its CU
    const bool actor = role == 0;
    const int cb = role * SH;                          // this role's first column of the stacked branch level
    const int m0 = tile * FT;
    const int r = tid >> 4, sub = tid & 15, m_row = m0 + r;
    const bool row_ok = m_row < M;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const float* img = p.cache_image;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];

#ifdef XRL_TILE_PROBE                                   // phase stamps: diagnostic builds only
    long long* dbg = p.dbg;
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - 1 - (unsigned)(dbg[15] == 77);   // dbg[15] = 77: stamp the actor
    long long tst[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) tst[i] = 0;
#define SSTAMP(k) do { if (dbg_me) tst[k] = clock64(); } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif
    SSTAMP(0);
    // ================= loads: gather (wave 7), parameter image, this role's 64 KB of W1 B-fragments (waves 0-3)
    float4 pf[PD];                                      // waves 0-3: output tile 4 role + wave of W1, all 16 k-chunks
    if (wave == 7) {
        const int m = m0 + (lane & 31);
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.f_rows) {
            if (m < M && lane < FT) {
                const float4* rec = reinterpret_cast<const float4*>(p.f_rows) + (size_t)m * 2;
                xr = rec[0]; sc = rec[1];
            }
        } else if (m < M && lane < FT) {
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
        if (lane < FT) { *reinterpret_cast<float4*>(xs + lane * 4) = xr; *reinterpret_cast<float4*>(rsc + lane * 4) = sc; }
    }
    float st_mean = 0.f, st_std = 1.f;
    if (p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    float4 imgv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < SI_FLOATS / 4) imgv = *reinterpret_cast<const float4*>(img + tid * 4);
    if (wave < 4) {
        const int t = 4 * role + wave;                                   // tile of the stacked 256-row W1
        const float* base = p.frag_image + ((size_t)t * (SH / 8) * 64 + lane) * 4;
#pragma unroll
        for (int q = 0; q < PD; ++q) pf[q] = *reinterpret_cast<const float4*>(base + frag_slot(q, t, SH / 8, 1) * 256);
    }
    if (tid < SI_FLOATS / 4) *reinterpret_cast<float4*>(pimg + tid * 4) = imgv;
    lds_barrier();                                                                                   // #0 gathered rows
    SSTAMP(1);
    const float4 xrow = *reinterpret_cast<const float4*>(xs + r * 4);
    const float4 rowsc = *reinterpret_cast<const float4*>(rsc + r * 4);
    const float g_act = rowsc.x, g_ret = rowsc.y, g_adv = rowsc.z, g_lp = rowsc.w;

    // ================= forward: first layer on the VALU (k-ordered fma chain == the MFMA result), both roles
    {
        float4 w0r[8], b0r[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) w0r[j] = *reinterpret_cast<const float4*>(pimg + SI_W0 + (sub * 8 + j) * 4);
        b0r[0] = *reinterpret_cast<const float4*>(pimg + SI_B0 + sub * 8);
        b0r[1] = *reinterpret_cast<const float4*>(pimg + SI_B0 + sub * 8 + 4);
        const float b0v[8] = {b0r[0].x, b0r[0].y, b0r[0].z, b0r[0].w, b0r[1].x, b0r[1].y, b0r[1].z, b0r[1].w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = __fmaf_rn(xrow.x, w0r[j].x, 0.f);
            acc = __fmaf_rn(xrow.y, w0r[j].y, acc);
            acc = __fmaf_rn(xrow.z, w0r[j].z, acc);
            acc = __fmaf_rn(xrow.w, w0r[j].w, acc);
            o[j] = act_apply_c<ACT>(acc + b0v[j]);
        }
        float* dst = h1 + r * SLD + sub * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    lds_barrier();                                                                                   // #1 h1
    SSTAMP(2);
    // ---- this role's branch layer 128 -> 128 on the matrix cores: wave w < 4 owns output columns [32 w, 32 w + 32)
    if (wave < 4) {
        const float* arow = h1 + li * SLD + 4 * lh;
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
        const int col = wave * 32 + li;
        const float bm = pimg[SI_BM + cb + col];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * SLD + col] = act_apply_c<ACT>(acc[rr] + bm);
        }
        // forward fragments consumed: the same registers take the BACKWARD section of the fragment copy for dH1 below (output
        // tile kt = wave, this role's 16 n-chunks q = 16 role + i; xrl_pack_mid_frags) -- a second stream that has the
        // head / loss / weight-gradient phases to arrive, instead of transposing the forward fragments through LDS
        // (csrc/ppo_fast.hip, csrc/ppo_wide.hip: measured there)
        {
            const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_image), 0, 2 * 2 * SH * SH * 4, 0x00020000);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                const int q = 16 * role + i;
                const su32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (2 * SH * SH + (wave * 32 + frag_slot(q, wave, 32, 2)) * 256) * 4, 0);
                pf[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                                                                   // #2 h2
    SSTAMP(3);

    // ================= head forward (VALU, 16 threads per row), this role's loss terms, head backward -- in registers
    // k-chunks q = sub + 16 i, i = 0, 1 (the 32 float4 chunks of this role's 128 columns)
    float4 a[2], wa[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float4*>(h2 + r * SLD + 4 * (sub + 16 * i));
        // merged head rows: 0, 1 = logits over columns [0, 128), 2 = value over columns [128, 256)
        wa[0][i] = *reinterpret_cast<const float4*>(pimg + SI_WH + (actor ? 0 : 2) * SI_LDH + cb + 4 * (sub + 16 * i));
        wa[1][i] = actor ? *reinterpret_cast<const float4*>(pimg + SI_WH + 1 * SI_LDH + 4 * (sub + 16 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float hv0, hv1;
    {
        float c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            c0 += a[i].x * wa[0][i].x + a[i].y * wa[0][i].y + a[i].z * wa[0][i].z + a[i].w * wa[0][i].w;
            c1 += a[i].x * wa[1][i].x + a[i].y * wa[1][i].y + a[i].z * wa[1][i].z + a[i].w * wa[1][i].w;
        }
        hv0 = srow16_sum(c0) + pimg[SI_BH + (actor ? 0 : 2)];
        hv1 = srow16_sum(c1) + pimg[SI_BH + 1];
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
            rowstat[0 * FT + r] = t_s; rowstat[1 * FT + r] = t_c; rowstat[2 * FT + r] = t_e; rowstat[3 * FT + r] = t_v; rowstat[4 * FT + r] = t_n;
        }
    }
    // dH2 = dZh . W_h, times act'(h2): this thread's two k-chunks
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
        *reinterpret_cast<float4*>(g2 + r * SLD + 4 * (sub + 16 * i)) = g;
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    SSTAMP(4);

    // ================= backward
    // ---- loss terms of this (tile, role): same tree as ppo_fast_kernel; the actor fills surrogate / entropy / clip count,
    //      the critic the value terms -- the partials reduction adds the rows of all workgroups
    if (wave == 7) {
        double acc_s = 0.0, acc_c = 0.0, acc_e = 0.0, acc_v = 0.0, acc_n = 0.0;
        if (lane < FT) { acc_s = rowstat[lane]; acc_c = rowstat[FT + lane]; acc_e = rowstat[2 * FT + lane]; acc_v = rowstat[3 * FT + lane]; acc_n = rowstat[4 * FT + lane]; }
        acc_s = wave_sum(acc_s); acc_c = wave_sum(acc_c); acc_e = wave_sum(acc_e); acc_v = wave_sum(acc_v); acc_n = wave_sum(acc_n);
        if (lane == 0) {
            double* q = p.partials + (size_t)blockIdx.x * 8;
            q[0] = acc_s; q[1] = acc_c; q[2] = acc_e; q[3] = acc_v; q[4] = acc_n; q[5] = 0; q[6] = 0; q[7] = 0;
        }
    }
    // ---- head weight / bias gradients and this role's branch-layer bias gradient: VALU reductions over the 32 rows
    if (tid < (actor ? 2 * SH : SH)) {
        const int j = tid >> 7, k = tid & (SH - 1);
        const float* hp = h2 + k;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 4 + j] * hp[rr * SLD];
        slab[(actor ? La.w_off + j * SH : Lc.w_off) + k] = acc;
    } else if (tid >= 4 * 64 && tid < 4 * 64 + SH) {
        const int t = tid - 4 * 64;
        float acc0 = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc0 += g2[rr * SLD + t];
        slab[L1.b_off + cb + t] = acc0;
        if (t < (actor ? 2 : 1)) {
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 4 + t];
            slab[actor ? La.b_off + t : Lc.b_off] = acc;
        }
    }
    SSTAMP(5);
    // ---- dW1[n][k] = sum_rows g2[row][n] * h1[row][k] for this role's 128 rows n: 4 x 4 tiles of 32 x 32, wave w owns
    //      n-tile (w & 3) and the k-tiles 2 (w >> 2), 2 (w >> 2) + 1; 16 chained MFMAs per tile, same k order as ppo_fast
    {
        const int nt = wave & 3, kt0 = 2 * (wave >> 2);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* arow = g2 + lh * SLD + nt * 32 + li;               // A[i = n][k = row]
        const float* brow = h1 + lh * SLD + kt0 * 32 + li;              // B[k = row][j]
#pragma unroll
        for (int s = 0; s < FT / 2; ++s) {
            const float av = arow[2 * s * SLD];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bv = brow[2 * s * SLD + t * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
        float* dW = slab + L1.w_off + (size_t)(cb + nt * 32) * SH;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)row * SH + (kt0 + t) * 32 + li] = acc[t][rr];
            }
    }
    SSTAMP(6);
    // ---- this role's part of dH1 = g2 . W1 (sum over its 128 rows n): wave kt < 4 owns output columns [32 kt, 32 kt + 32);
    //      B operand = the backward fragments requested after the forward layer (n-chunks ascending: the order it always had)
    if (wave < 4) {
        const int k_out = wave * 32 + li;
        const float* arow = g2 + li * SLD + 4 * lh;
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
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            xb[row * SLD + k_out] = acc[rr] * act_grad_c<ACT>(h1[row * SLD + k_out]);
        }
    }
    lds_barrier();                                                                                   // #5 g1 (this role's part)
    SSTAMP(7);
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c] -- the actor's part into the slab's first-layer
    //      region, the critic's into the fold region behind the parameters (the reduction adds it onto the same columns)
    {
        float* dst = actor ? slab : slab + p.l0_fold_off;
        const int w_at = actor ? L0.w_off : 0, b_at = actor ? L0.b_off : 4 * SH;
        const int c = tid >> 2, k = tid & 3;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += xb[rr * SLD + c] * xs[rr * 4 + k];
        dst[w_at + tid] = acc;
        if (tid < SH) {
            float accb = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) accb += xb[rr * SLD + tid];
            dst[b_at + tid] = accb;
        }
    }
    SSTAMP(8);
#ifdef XRL_TILE_PROBE
    if (dbg_me) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dbg[i] = tst[i] - tst[0];
    }
#endif
#undef SSTAMP
}

extern bool g_fast_enabled_ppo;
bool ppo_fast_eligible(const xrl_ppo_fused_t& p);

// the role-split form needs what ppo_fast needs, plus the fragment image, a fold region and the first layer at the front
bool ppo_split_eligible(const xrl_ppo_fused_t& p) {
    if (!ppo_fast_eligible(p) || !p.frag_image || p.l0_fold_off <= 0) return false;
    const xrl_fused_layer_t& L0 = p.layers[0];
    return L0.w_off == 0 && L0.b_off == 4 * SH && (p.l0_fold_off & 3) == 0 && p.l0_fold_off + SP_FOLD <= p.slab_stride;
}

int launch_ppo_split(const xrl_ppo_fused_t& p0, hipStream_t stream) {
    static const int only = getenv("XRL_SPLIT_ONLY_ROLE") ? atoi(getenv("XRL_SPLIT_ONLY_ROLE")) + 1 : 0;
    xrl_ppo_fused_t p = p0;
    p.pad0 = only;
    const int n_tiles = (p.M + FT - 1) / FT;
    XRL_ACT_DISPATCH(p.layers[0].act,
        hipLaunchKernelGGL(ppo_split_kernel<ACT>, dim3(2 * n_tiles), dim3(FUSED_THREADS), SP_LDS_BYTES, stream, p);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

int init_ppo_split() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_split_kernel<XRL_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_split_kernel<XRL_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_split_kernel<XRL_ACT_LEAKY_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_split_kernel<XRL_ACT_TANH>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_split_kernel<XRL_ACT_SIGMOID>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES));
    return XRL_OK;
}

}  // namespace xrl

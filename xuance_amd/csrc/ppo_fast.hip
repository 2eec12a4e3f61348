// Shape-specialised fused PPO minibatch for the actor-critic 4-128-{128-2, 128-1} (BASELINE configs[0..1]): same
// contract, arithmetic and bit-identical results as ppo_fused_kernel (ppo_fused.hip, the any-shape kernel) with every
// extent a compile-time constant.  What that buys (see rollout_fast.hip for the measurements behind each rule):
//   * no LDS parameter cache -- first-layer / head weights and biases sit in the registers of the threads that use them;
//   * both 128 KB weight streams (W1 for the forward, W1^T for backward-data) are B-fragment register prefetches that
//     stay in flight across LDS-only barriers; the second one is issued as soon as the first has been consumed and has
//     the loss / weight-gradient phases to arrive;
//   * every loop is unrolled, so each phase issues its LDS reads back to back instead of one latency per access;
//   * the 16 threads of a row all hold the row's logits after the DPP reduction, so the loss and the head backward need
//     no broadcast and no extra barrier; h2 chunks stay in registers between the head forward and backward.
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward).
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"

namespace xrl {

constexpr int PH = 128;                      // hidden width
constexpr int PLD1 = PH + 4;                 // row stride of 128-wide levels
constexpr int PLD2 = 2 * PH + 4;             // row stride of the stacked actor|critic level
// packed image layout (pack_rollout_cache_kernel) for 4-128-256-{2|1}
constexpr int PI_W0 = 0, PI_B0 = 4 * PH, PI_BM = PI_B0 + PH, PI_WH = PI_BM + 2 * PH, PI_LDH = 2 * PH + 4, PI_BH = PI_WH + 3 * PI_LDH;
constexpr int PI_FLOATS = PI_BH + 4;          // 1680: first layer | biases | merged heads
constexpr int PF_LDS_FLOATS = FT * (2 * PLD1 + 2 * PLD2) + NW * 32 * 33 + 3 * FT * 4 + PI_FLOATS;
constexpr int PF_LDS_BYTES = PF_LDS_FLOATS * 4 + FT * 5 * 8;

typedef unsigned fu32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float pdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// all-reduce over a row of 16 lanes; bit-identical to the xor butterfly 8,4,2,1 (see rollout_fast.hip)
__device__ __forceinline__ float row16_sum(float v) {
    v += pdpp<0x128>(v); v += pdpp<0x124>(v); v += pdpp<0x122>(v); v += pdpp<0x121>(v);
    return v;
}

template <int ACT>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_fast_kernel(xrl_ppo_fused_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                   // [32][132] first hidden level
    float* h2 = h1 + FT * PLD1;                        // [32][260] actor | critic hidden level
    float* g2 = h2 + FT * PLD2;                        // [32][260] dLoss/d(pre-activation of h2)
    float* g1 = g2 + FT * PLD2;                        // [32][132] dLoss/d(pre-activation of h1)
    float* red = g1 + FT * PLD1;                       // [8][32][33] split-K partial tiles
    float* xs = red + NW * 32 * 33;                    // [32][4] gathered observations
    float* dzh = xs + FT * 4;                          // [32][4] dLoss/d(logits, value)
    float* rsc = dzh + FT * 4;                         // [32][4] gathered act | ret | adv | old_logp
    float* pimg = rsc + FT * 4;                        // [PI_FLOATS] copy of the packed small-parameter image
    double* rowstat = reinterpret_cast<double*>(pimg + PI_FLOATS);   // [5][32] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_fused_t)>();
    constexpr int D = 4, A = 2;
    const int tid = threadIdx.x, M = p.M;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * FT;
    const int r = tid >> 4, sub = tid & 15, m_row = m0 + r;
    const bool row_ok = m_row < M;
    float* slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
    const float* img = p.cache_image;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
#ifdef XRL_TILE_PROBE                                   // phase stamps cost 20 VGPRs: diagnostic builds only
    long long* dbg = p.dbg;
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - 1;
    long long tst[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) tst[i] = 0;
#define QSTAMP(k) do { if (dbg_me) tst[k] = clock64(); } while (0)
#else
#define QSTAMP(k) do { } while (0)
#endif
    QSTAMP(0);

    // ================= loads.  A CU streams weights from L2 at only ~10 B/clk (64 outstanding 64-byte misses per L1 and
    // ~350 cycles of L2 latency), so the two 128 KB streams take as long as all the matrix-core work of the tile and must
    // flow from the first cycle; the gather (a dependent chain index -> record) would queue behind them in every wave, so
    // ONE wave (7) does it for all 32 rows before starting its own share of the streams and hands the rows over in LDS.
    float4 pf[PD];                                      // B fragments of W1: output tile `wave` (rows 32 w .. 32 w + 31), all 16 k-chunks
    if (wave == 7) {
        const int m = m0 + (lane & 31);
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.f_rows) {                                                 // records gathered for the whole update phase
            if (m < M && lane < FT) {
                const float4* rec = reinterpret_cast<const float4*>(p.f_rows) + (size_t)m * 2;
                xr = rec[0]; sc = rec[1];
            }
        } else if (m < M && lane < FT) {
            const int64_t fl = p.idx[m];
            const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);   // (env, t) = divmod(idx, T); field[t][env]
            const size_t src = (size_t)t * p.n_envs + env;
            if (p.f_packed) {                                           // one 32-byte record per row
                const float4* rec = reinterpret_cast<const float4*>(p.f_packed) + src * 2;
                xr = rec[0]; sc = rec[1];
            } else {
                xr = *reinterpret_cast<const float4*>(p.f_obs + src * D);
                sc = make_float4(p.f_act[src], p.f_ret[src], p.f_adv[src], p.f_logp[src]);
            }
        }
        if (lane < FT) { *reinterpret_cast<float4*>(xs + lane * 4) = xr; *reinterpret_cast<float4*>(rsc + lane * 4) = sc; }
    }
    // small loads first: the parameter image (first layer, biases, heads: 6.7 KB, handed over through LDS) and the
    // advantage statistics must not queue behind the stream in this wave's in-order vmcnt
    float st_mean = 0.f, st_std = 1.f;
    if (p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    float4 imgv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < PI_FLOATS / 4) imgv = *reinterpret_cast<const float4*>(img + tid * 4);
    // First weight stream: the forward fragments.  Backward-data needs W1 transposed: a second 128 KB stream (the fragment
    // copy's backward section) is requested right after the forward layer, see there; the per-CU stream rate (~10 B/clk)
    // makes it as long as all the matrix-core work of the tile, but nothing waits for it until dH1.
    {
        // one load instruction per chunk whatever the source layout (two code paths assigning pf[] make hipcc split every
        // 16-byte load into four 4-byte ones: 4x the instructions and 4x the L1 traffic)
        const bool fo = p.frag_image != nullptr;
        const float* base = fo ? p.frag_image + ((size_t)wave * (PH / 8) * 64 + lane) * 4
                               : p.params + L1.w_off + (size_t)(wave * 32 + li) * PH + 4 * lh;
#pragma unroll
        for (int q = 0; q < PD; ++q)
            pf[q] = *reinterpret_cast<const float4*>(base + (fo ? frag_slot(q, wave, PH / 8, 1) * 256 : q * 8));
    }
    // (vmcnt retires in order: the image was LOADED before the stream so that it can be stored without waiting for it)
    if (tid < PI_FLOATS / 4) *reinterpret_cast<float4*>(pimg + tid * 4) = imgv;
    lds_barrier();                                                                                   // #0 gathered rows
    QSTAMP(9);
    const float4 xrow = *reinterpret_cast<const float4*>(xs + r * 4);
    const float4 rowsc = *reinterpret_cast<const float4*>(rsc + r * 4);
    const float g_act = rowsc.x, g_ret = rowsc.y, g_adv = rowsc.z, g_lp = rowsc.w;

    // ================= forward: first layer on the VALU (k-ordered fma chain == the MFMA result)
    {
        float4 w0r[8], b0r[2];                          // thread (r, sub) -> columns [8 sub, 8 sub + 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) w0r[j] = *reinterpret_cast<const float4*>(pimg + PI_W0 + (sub * 8 + j) * 4);
        b0r[0] = *reinterpret_cast<const float4*>(pimg + PI_B0 + sub * 8);
        b0r[1] = *reinterpret_cast<const float4*>(pimg + PI_B0 + sub * 8 + 4);
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
        float* dst = h1 + r * PLD1 + sub * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    lds_barrier();                                                                                   // #1 h1
    QSTAMP(1);
    // ---- branch layer 128 -> 256 on the matrix cores: wave w owns output columns [32 w, 32 w + 32)
    {
        const float* arow = h1 + li * PLD1 + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {                // A fragments in two batches of 8 (register budget: both streams live)
            float4 af[PD / 2];
#pragma unroll
            for (int q = 0; q < PD / 2; ++q) af[q] = *reinterpret_cast<const float4*>(arow + (hq * 8 + q) * 8);
#pragma unroll
            for (int q = 0; q < PD / 2; ++q) { MFMA4(af[q], pf[hq * 8 + q], acc) }
        }
        const int col = wave * 32 + li;
        const float bm = pimg[PI_BM + col];             // branch-layer bias of this lane's output column
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * PLD2 + col] = act_apply_c<ACT>(acc[rr] + bm);
        }
        // The forward fragments are consumed: the same registers take this wave's share of the BACKWARD section of the
        // fragment copy (W1 with the reduction index n on the fragment's k axis; output tile kt = wave / 2, n-chunks
        // q = half, half + 2, ..: xrl_pack_mid_frags), needed at dH1.  It has the head / loss / weight-gradient phases
        // (~14 k cycles, no global loads) to arrive -- 128 KB per workgroup at the ~10 B/clk a CU pulls from L2.  Through a
        // buffer descriptor: one VGPR of per-lane offset for all 16 loads, the chunk's place in the scalar offset.
        // (Until round 2 the forward fragments were transposed through LDS in four 32 KB stages instead: 17 k cycles for
        // 8 k of matrix work; measured on the 256-wide sibling, csrc/ppo_wide.hip.)
        {
            const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_image), 0, 2 * 2 * PH * PH * 4, 0x00020000);
            const int kt = wave >> 1, half = wave & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                const int q = half + 2 * i;
                const fu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (2 * PH * PH + (kt * 32 + frag_slot(q, kt, 32, 2)) * 256) * 4, 0);
                pf[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                                                                   // #2 h2
    QSTAMP(2);

    // ================= heads forward (VALU, 16 threads per row), loss, heads backward -- all in registers
    // k-chunks q = sub + 16 i; i = 0,1 lie in the actor half (rows 0,1 of the merged head), i = 2,3 in the critic half
    float4 a[4], wa[2][2], wc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4*>(h2 + r * PLD2 + 4 * (sub + 16 * i));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        wa[0][i] = *reinterpret_cast<const float4*>(pimg + PI_WH + 0 * PI_LDH + 4 * (sub + 16 * i));
        wa[1][i] = *reinterpret_cast<const float4*>(pimg + PI_WH + 1 * PI_LDH + 4 * (sub + 16 * i));
        wc[i] = *reinterpret_cast<const float4*>(pimg + PI_WH + 2 * PI_LDH + 4 * (sub + 16 * (i + 2)));
    }
    const float bh0 = pimg[PI_BH + 0], bh1 = pimg[PI_BH + 1], bh2 = pimg[PI_BH + 2];
    float hv0, hv1, hv2;
    {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            c0 += a[i].x * wa[0][i].x + a[i].y * wa[0][i].y + a[i].z * wa[0][i].z + a[i].w * wa[0][i].w;
            c1 += a[i].x * wa[1][i].x + a[i].y * wa[1][i].y + a[i].z * wa[1][i].z + a[i].w * wa[1][i].w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            c2 += a[i + 2].x * wc[i].x + a[i + 2].y * wc[i].y + a[i + 2].z * wc[i].z + a[i + 2].w * wc[i].w;
        hv0 = row16_sum(c0) + bh0; hv1 = row16_sum(c1) + bh1; hv2 = row16_sum(c2) + bh2;
    }
    float dz0 = 0.f, dz1 = 0.f, dz2 = 0.f;
    {
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        if (row_ok) {
            float adv = g_adv;
            asm volatile("" : "+v"(st_std));     // keeps hipcc from consuming the statistics (and waiting) at the top
            if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);           // memory_tools.py:281-282
            const float invM = 1.f / (float)M;
            const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
            const int act = (int)g_act;
            const float o[2] = {hv0, hv1};
            const float v = hv2;
            float mx = o[0];
            mx = fmaxf(mx, o[1]);
            float se = 0.f;
            se += expf(o[0] - mx); se += expf(o[1] - mx);
            const float lse = mx + logf(se);
            const float logp = (act == 0 ? o[0] : o[1]) - lse;
            float ent = 0.f;
#pragma unroll
            for (int j = 0; j < A; ++j) { const float l = o[j] - lse; ent -= expf(l) * l; }
            const Surrogate s = surrogate(logp, g_lp, adv, lo, hi, invM);
            const float ce = p.ent_coef * invM;
            float dq[2];
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float l = o[j] - lse, pj = expf(l);
                dq[j] = s.dlogp * ((j == act ? 1.f : 0.f) - pj) + ce * pj * (l + ent);
            }
            const float dv = v - g_ret;
            dz0 = dq[0]; dz1 = dq[1]; dz2 = p.vf_coef * 2.f * dv * invM;
            t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_c = (double)dv * dv; t_e = ent; t_v = v;
            if (p.diag && sub == 0) {
                const int m = m_row;
                p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
            }
        }
        if (sub == 0) {
            dzh[r * 4 + 0] = dz0; dzh[r * 4 + 1] = dz1; dzh[r * 4 + 2] = dz2;
            rowstat[0 * FT + r] = t_s; rowstat[1 * FT + r] = t_c; rowstat[2 * FT + r] = t_e; rowstat[3 * FT + r] = t_v; rowstat[4 * FT + r] = t_n;
        }
    }
    // dH2 = dZh . W_h, times act'(h2): this thread's four k-chunks
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 g;
        g.x = (dz0 * wa[0][i].x + dz1 * wa[1][i].x) * act_grad_c<ACT>(a[i].x);
        g.y = (dz0 * wa[0][i].y + dz1 * wa[1][i].y) * act_grad_c<ACT>(a[i].y);
        g.z = (dz0 * wa[0][i].z + dz1 * wa[1][i].z) * act_grad_c<ACT>(a[i].z);
        g.w = (dz0 * wa[0][i].w + dz1 * wa[1][i].w) * act_grad_c<ACT>(a[i].w);
        *reinterpret_cast<float4*>(g2 + r * PLD2 + 4 * (sub + 16 * i)) = g;
        float4 c;
        c.x = (dz2 * wc[i].x) * act_grad_c<ACT>(a[i + 2].x);
        c.y = (dz2 * wc[i].y) * act_grad_c<ACT>(a[i + 2].y);
        c.z = (dz2 * wc[i].z) * act_grad_c<ACT>(a[i + 2].z);
        c.w = (dz2 * wc[i].w) * act_grad_c<ACT>(a[i + 2].w);
        *reinterpret_cast<float4*>(g2 + r * PLD2 + 4 * (sub + 16 * (i + 2))) = c;
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    QSTAMP(3);

    // ================= backward
    // ---- loss terms of the tile: same reduction tree as the any-shape kernel (rows on lanes 0..31 of one wave)
    if (wave == 7) {
        double acc_s = 0.0, acc_c = 0.0, acc_e = 0.0, acc_v = 0.0, acc_n = 0.0;
        if (lane < FT) { acc_s = rowstat[lane]; acc_c = rowstat[FT + lane]; acc_e = rowstat[2 * FT + lane]; acc_v = rowstat[3 * FT + lane]; acc_n = rowstat[4 * FT + lane]; }
        acc_s = wave_sum(acc_s); acc_c = wave_sum(acc_c); acc_e = wave_sum(acc_e); acc_v = wave_sum(acc_v); acc_n = wave_sum(acc_n);
        if (lane == 0) {
            double* q = p.partials + (size_t)blockIdx.x * 8;
            q[0] = acc_s; q[1] = acc_c; q[2] = acc_e; q[3] = acc_v; q[4] = acc_n; q[5] = 0; q[6] = 0; q[7] = 0;
        }
    }
    // ---- head weight / bias gradients (waves 0-5) and the branch-layer bias gradient (waves 6-7): VALU reductions over
    //      the 32 rows, all LDS reads of a thread issued back to back
    if (wave < 6) {
        const int j = tid >> 7, k = tid & (PH - 1);
        const float* hp = h2 + (j == 2 ? PH : 0) + k;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 4 + j] * hp[rr * PLD2];
        slab[(j == 2 ? Lc.w_off : La.w_off + j * PH) + k] = acc;
    } else {
        const int t = tid - 6 * 64;                                      // 0..127
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) { acc0 += g2[rr * PLD2 + t]; acc1 += g2[rr * PLD2 + PH + t]; }
        slab[L1.b_off + t] = acc0; slab[L1.b_off + PH + t] = acc1;
        if (t < 3) {
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 4 + t];
            slab[t == 2 ? Lc.b_off : La.b_off + t] = acc;
        }
    }
    QSTAMP(4);
    // ---- dW1[n][k] = sum_rows g2[row][n] * h1[row][k]: wave w owns rows n in [32 w, 32 w + 32), 4 column tiles
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* arow = g2 + lh * PLD2 + wave * 32 + li;            // A[i = n][k = row]
        const float* brow = h1 + lh * PLD1 + li;                        // B[k = row][j]
#pragma unroll
        for (int s = 0; s < FT / 2; ++s) {
            const float av = arow[2 * s * PLD2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bv = brow[2 * s * PLD1 + t * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
        float* dW = slab + L1.w_off + (size_t)(wave * 32) * PH;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)row * PH + t * 32 + li] = acc[t][rr];
            }
    }
    QSTAMP(5);
    // ---- dH1 = g2 . W1: output tile kt = wave / 2 (columns 32 kt ..), n-half = wave & 1 (chunks q = half, half + 2, ..,
    //      ascending: the same summation order as ever), partial tiles meet in `red`.  The B operand comes from the
    //      backward fragments requested after the forward layer.
    {
        const int kt = wave >> 1, half = wave & 1;
        const float* arow = g2 + li * PLD2 + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            float4 af[PD / 2];
#pragma unroll
            for (int i = 0; i < PD / 2; ++i) af[i] = *reinterpret_cast<const float4*>(arow + (half + 2 * (hq * 8 + i)) * 8);
#pragma unroll
            for (int i = 0; i < PD / 2; ++i) { MFMA4(af[i], pf[hq * 8 + i], acc) }
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            red[(wave * 32 + row) * 33 + li] = acc[rr];
        }
        (void)kt;
    }
    lds_barrier();                                                                                   // #4 red
    QSTAMP(6);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + j * FUSED_THREADS;
        const int tl = i >> 10, row = (i >> 5) & 31, c = i & 31, col = tl * 32 + c;
        float v = 0.f;
        v += red[((tl * 2 + 0) * 32 + row) * 33 + c];
        v += red[((tl * 2 + 1) * 32 + row) * 33 + c];
        g1[row * PLD1 + col] = v * act_grad_c<ACT>(h1[row * PLD1 + col]);
    }
    lds_barrier();                                                                                   // #5 g1
    QSTAMP(7);
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c]
    {
        const int c = tid >> 2, k = tid & 3;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += g1[rr * PLD1 + c] * xs[rr * 4 + k];
        slab[L0.w_off + tid] = acc;
        if (tid < PH) {
            float accb = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) accb += g1[rr * PLD1 + tid];
            slab[L0.b_off + tid] = accb;
        }
    }
    QSTAMP(8);
#ifdef XRL_TILE_PROBE
    if (dbg_me) {
#pragma unroll
        for (int i = 0; i < 10; ++i) dbg[i] = tst[i] - tst[0];
        dbg[15] = 10;
    }
#endif
#undef QSTAMP
}

extern bool g_fast_enabled_ppo;
bool g_fast_enabled_ppo = true;

bool ppo_fast_eligible(const xrl_ppo_fused_t& p) {
    if (!g_fast_enabled_ppo) return false;
    if (p.D != 4 || p.A != 2 || p.n_layers != 4 || p.n_head_layers != 2 || p.n_levels != 4 || !p.frag_image) return false;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &Ha = p.layers[2], &Hc = p.layers[3];
    if (p.level_width[1] != PH || p.level_width[2] != 2 * PH || p.level_width[3] != 3) return false;
    if (L0.K != 4 || L0.N != PH || L0.in_level != 0 || L0.out_level != 1 || L0.out_off != 0) return false;
    if (L1.K != PH || L1.N != 2 * PH || L1.in_level != 1 || L1.in_off != 0 || L1.out_level != 2 || L1.out_off != 0) return false;
    if (L1.act != L0.act) return false;
    if (Ha.K != PH || Ha.N != 2 || Ha.in_level != 2 || Ha.in_off != 0 || Ha.out_level != 3 || Ha.out_off != 0 || Ha.act != XRL_ACT_NONE) return false;
    if (Hc.K != PH || Hc.N != 1 || Hc.in_level != 2 || Hc.in_off != PH || Hc.out_level != 3 || Hc.out_off != 2 || Hc.act != XRL_ACT_NONE) return false;
    if ((reinterpret_cast<uintptr_t>(p.params + L1.w_off) & 15) || (reinterpret_cast<uintptr_t>(p.params_t + L1.w_off) & 15)) return false;
    return true;
}

int launch_ppo_fast(const xrl_ppo_fused_t& p, hipStream_t stream) {
    const int n_tiles = (p.M + FT - 1) / FT;
    XRL_ACT_DISPATCH(p.layers[0].act,
        hipLaunchKernelGGL(ppo_fast_kernel<ACT>, dim3(n_tiles), dim3(FUSED_THREADS), PF_LDS_BYTES, stream, p);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

int init_ppo_fast() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fast_kernel<XRL_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fast_kernel<XRL_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fast_kernel<XRL_ACT_LEAKY_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fast_kernel<XRL_ACT_TANH>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_fast_kernel<XRL_ACT_SIGMOID>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
    return XRL_OK;
}

}  // namespace xrl

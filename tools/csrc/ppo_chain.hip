// PPO minibatch for the shared-trunk family D -> 128 -> {128 -> A | 128 -> 1} with a categorical head, A <= 4, on 64-row tiles:
// the (tile, role) workgroups of csrc/ppo_trunk.hip with the FORWARD and the backward-data product computed as register chains in
// the transposed form of the rollout kernels (csrc/rollout_actor.hip).  Round 4: the per-wave stamps of ppo_trunk_kernel showed 20 k
// of a workgroup's 44.7 k cycles outside the matrix pipe -- LDS hand-overs of every level, barrier skew around the three MFMA
// phases, two 64 KB weight-fragment streams.  Here:
//   * ONE 64 KB weight stream per workgroup: this role's 128 x 128 block of the branch layer goes to LDS once, XOR-swizzled so that
//     it can be read conflict-free in BOTH orientations (forward: A[m = unit][k], 16-byte reads; backward-data: A[m = k][unit]);
//   * waves 0..3 (chain waves) own 16 rows each: first layer, branch layer, head, loss, head backward and g2 stay in registers --
//     v_mfma_f32_16x16x4_f32 computes D[unit][row]; the D layout of a 16-unit tile is the B-operand layout of the next product
//     over those units -- and the backward-data product dH1^T = W1^T . g2^T chains from the same registers;
//   * h1 and g2 go to LDS once (for the weight gradient, which contracts over ROWS); waves 4..7 form dW1 (ppo_trunk_kernel's loop,
//     row order unchanged) WHILE the chain waves form dH1 -- both on the matrix pipe, no barrier between them;
//   * vector sums over rows (head weights / biases, branch bias) are 16-lane DPP reductions of the chain waves' registers, four
//     partials per element met in LDS.
// MEASURED (round 4, 8 192-row minibatch, tools/probe_pair_phases.py): 31.1 us per launch alone against 25.3 us for ppo_trunk_kernel
// -- NOT the default (config.use_chain_update).  Cycles of a workgroup: rows + small parameters 6.8 k, weight block + first layer
// 5.0 k (the first layer's operand loads retire behind the 64 KB weight block: loads return in order), branch layer + head + loss +
// g2 + the DPP row sums 17.3 k (8.2 k of MFMA; ~1 800 vector instructions, 390 of them DPP moves the compiler does not fold into
// the adds), dW1 || dH1 24.1 k (16.4 k of MFMA; the helper waves' 64 KB of dW1 stores in one burst at the end), tail 3.3 k = 56.5 k
// against 44.6 k.  What the form buys -- one weight stream, no hand-over of h1 / h2 through LDS in the forward -- is spent on the
// row reductions that the [row][unit] layout of ppo_trunk_kernel gets from plain LDS column loops.
// Gradient slabs, loss partials, fold region: the layout of ppo_trunk_kernel (xrl_reduce_adam does not know which kernel wrote them).
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward),
// distributions.py:128-153 (CategoricalDistribution), actor_head.py:14-43.
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"

namespace xrl {

typedef float cf32x4 __attribute__((ext_vector_type(4)));

constexpr int CH = 128;                      // hidden width
constexpr int CLD = CH + 4;                  // row stride of the [row][unit] levels in LDS
constexpr int CPT = 64;                      // rows per workgroup
constexpr int CXLD = 28, CDMAX = 24;         // gathered observations: row stride / width limit
constexpr int CAMAX = 4;                     // head width limit of this kernel
constexpr int CKC = CDMAX / 4;               // k-steps of the first layer (4 observation components each)

struct ChainLds {
    static constexpr int W1S = 0, H1 = W1S + CH * CH, G2 = H1 + CPT * CLD, XS = G2 + CPT * CLD, RSC = XS + CPT * CXLD,
                         B0 = RSC + CPT * 4, BM = B0 + CH, WHS = BM + CH, BH = WHS + CAMAX * CH, PARTW = BH + 4,
                         PARTB = PARTW + 4 * CAMAX * CH, PARTHB = PARTB + 4 * CH, SRC = PARTHB + 4 * 4, FLOATS = SRC + CPT;
    static constexpr int BYTES = FLOATS * 4 + CPT * 5 * 8;
};
static_assert(ChainLds::BYTES <= 160 * 1024, "tile does not fit the LDS of a CU");
static_assert((ChainLds::FLOATS % 2) == 0, "row statistics are doubles");

template <int CTRL>
__device__ __forceinline__ float cdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (= the 16 rows of a chain wave's tile); every lane gets it
__device__ __forceinline__ float crow16_sum(float v) {
    v += cdpp<0x128>(v); v += cdpp<0x124>(v); v += cdpp<0x122>(v); v += cdpp<0x121>(v);
    return v;
}

#define CMFMA(a, b, acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), acc, 0, 0, 0)

// rows [r0, r0 + 32) of column c of g1 times NQ2 float2 chunks of the observations (ppo_trunk.hip: dw0_rows)
template <int NQ2>
__device__ __forceinline__ void cdw0_rows(const float* gcol, const float* xin, float (&acc)[CDMAX / 2], float& ab) {
#pragma unroll 4
    for (int rr = 0; rr < CPT / 2; ++rr) {
        const float g = gcol[rr * CLD];
        ab += g;
#pragma unroll
        for (int q = 0; q < NQ2; ++q) {
            const float2 x = *reinterpret_cast<const float2*>(xin + rr * CXLD + 2 * q);
            acc[2 * q] += g * x.x; acc[2 * q + 1] += g * x.y;
        }
    }
}

template <int ACT, int DS, int AS>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_chain_kernel(xrl_ppo_fused_t p) {
    using L = ChainLds;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1s = lds + L::W1S;                         // [128 units][32 chunks of 4, chunk q of unit u at slot q ^ (u & 15)]
    float* h1 = lds + L::H1;                           // [64][132]
    float* g2 = lds + L::G2;                           // [64][132]; later g1 (dLoss/d pre-activation of h1, this role's part)
    float* xb = g2;
    float* xs = lds + L::XS;                           // [64][28] gathered observations, zero beyond D
    float* rsc = lds + L::RSC;                         // [64][4] act | ret | adv | old_logp
    float* b0s = lds + L::B0;
    float* bms = lds + L::BM;                          // this role's branch bias
    float* whs = lds + L::WHS;                         // [4][128] this role's head rows (rows >= nout zero)
    float* bhs = lds + L::BH;
    float* partw = lds + L::PARTW;                     // [4 chain waves][4][128] head-weight gradient partials
    float* partb = lds + L::PARTB;                     // [4][128] branch-bias gradient partials
    float* parthb = lds + L::PARTHB;                   // [4][4] head-bias gradient partials
    int* srcs = reinterpret_cast<int*>(lds + L::SRC);
    double* rowstat = reinterpret_cast<double*>(lds + L::FLOATS);       // [5][64] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_fused_t)>();
    const int tid = threadIdx.x, M = p.M, D = DS ? DS : p.D, A = AS ? AS : p.A;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;           // 16x16x4 MFMA lane coordinates
    const int li = lane & 31, lh = lane >> 5;          // 32x32x2
    const int tile = blockIdx.x >> 1, role = blockIdx.x & 1;
    const bool actor = role == 0;
    const int nout = actor ? A : 1;
    const int cb = role * CH;
    const int m0 = tile * CPT;
    const bool chain = wave < 4;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
    const xrl_fused_layer_t& Lh = actor ? La : Lc;

    long long* dbg = p.dbg;
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - 1;
#define CSTAMP(k) do { if (dbg_me) dbg[k] = clock64(); } while (0)
    CSTAMP(0);

    // ================= loads: rows and small parameters first (a wave's loads retire in order), then the 64 KB weight block
    const bool records = D == 4 && (p.f_rows || p.f_packed);
    if (records) {                                     // 32-byte records obs[4] | act | ret | adv | old_logp: one wave
        if (wave == 7) {
            const int m = m0 + lane;
            float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M) {
                size_t at = (size_t)m;
                if (!p.f_rows) {
                    const int64_t fl = p.idx[m];
                    const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
                    at = (size_t)t * p.n_envs + env;
                }
                const float4* rec = reinterpret_cast<const float4*>(p.f_rows ? p.f_rows : p.f_packed) + at * 2;
                xr = rec[0]; sc = rec[1];
            }
            *reinterpret_cast<float4*>(xs + lane * CXLD) = xr;
            *reinterpret_cast<float4*>(xs + lane * CXLD + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(rsc + lane * 4) = sc;
        }
    } else if (tid < CPT) {                            // buffer row of each minibatch row (env-major flat index, memory_tools.py:270)
        const int m = m0 + tid;
        int src = -1;
        if (m < M) {
            const int64_t fl = p.idx[m];
            const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
            src = t * p.n_envs + env;
        }
        srcs[tid] = src;
    }
    // (every load below is UNCONDITIONAL -- clamped address, value selected afterwards: a load under a branch makes hipcc wait for all
    //  loads in flight at its first use, and the 64 KB weight block is in flight behind these)
    float st_mean = 0.f, st_std = 1.f;
    {
        const float* sp = p.stats ? p.stats : p.params;
        const float a = sp[0], b = sp[1];
        if (p.stats) { st_mean = a; st_std = b; }
    }
    // small parameters: thread -> (b0 | this role's branch bias | head bias) and one head-row element
    const int sm_at = tid < CH ? L0.b_off + tid : tid < 2 * CH ? L1.b_off + cb + tid - CH : Lh.b_off + min(tid - 2 * CH, nout - 1);
    const float smv = p.params[sm_at];
    const int hj = tid >> 7, hk = tid & (CH - 1);                         // 512 threads = 4 head rows x 128
    const float whv = p.params[Lh.w_off + min(hj, nout - 1) * CH + hk];
    // first-layer weights as MFMA A operands: A[m = unit 16 t + cl][k = component 4 c + g] (every wave loads them; the chain waves use them)
    constexpr int KC = DS ? (DS + 3) / 4 : CKC;
    float w0r[8][KC];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int d = 4 * c + g;
            const float v = p.params[L0.w_off + (size_t)(16 * t + cl) * D + min(d, D - 1)];
            w0r[t][c] = d < D ? v : 0.f;
        }
    // this role's 128 rows of the branch layer W1[256][128]: 8 float4 per thread (float4 f = tid + 512 j: unit f / 32, chunk f % 32)
    float4 w1v[8];
    {
        const float4* w1g = reinterpret_cast<const float4*>(p.params + L1.w_off + (size_t)cb * CH);
#pragma unroll
        for (int j = 0; j < 8; ++j) w1v[j] = w1g[tid + FUSED_THREADS * j];
    }
    if (tid < CH) b0s[tid] = smv;
    else if (tid < 2 * CH) bms[tid - CH] = smv;
    else if (tid < 2 * CH + 4) bhs[tid - 2 * CH] = tid - 2 * CH < nout ? smv : 0.f;
    whs[tid] = hj < nout ? whv : 0.f;
    if (!records) {
        lds_barrier();                                                                               // (srcs)
        for (int e = tid; e < CPT * CXLD; e += FUSED_THREADS) {
            const int rr = e / CXLD, k = e - rr * CXLD, src = srcs[rr];
            xs[e] = (k < D && src >= 0) ? p.f_obs[(size_t)src * D + k] : 0.f;
        }
        for (int e = tid; e < CPT * 4; e += FUSED_THREADS) {
            const int rr = e >> 2, k = e & 3, src = srcs[rr];
            float v = 0.f;
            if (src >= 0) v = k == 0 ? p.f_act[src] : k == 1 ? p.f_ret[src] : k == 2 ? p.f_adv[src] : p.f_logp[src];
            rsc[e] = v;
        }
    }
    lds_barrier();                                                                                   // #0 rows, small parameters
    CSTAMP(1);

    // ================= chain waves: first layer (transposed: D[unit][row]); everybody: the weight block into LDS
    const int row = 16 * wave + cl;                    // (chain waves) this lane's row of the tile
    cf32x4 h1T[8];                                     // h1T[t][i] = h1[row][unit 16 t + 4 g + i]
    if (chain) {
        float xT[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) xT[c] = xs[row * CXLD + 4 * c + g];                              // B[k = component][n = row]
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 bb = *reinterpret_cast<const float4*>(&b0s[16 * t + 4 * g]);
            cf32x4 acc = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int c = 0; c < KC; ++c) CMFMA(w0r[t][c], xT[c], acc);
            cf32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = act_apply_c<ACT>(acc[i]);
            h1T[t] = o;
            *reinterpret_cast<float4*>(h1 + row * CLD + 16 * t + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = tid + FUSED_THREADS * j, u = f >> 5, q = f & 31;
        *reinterpret_cast<float4*>(w1s + u * CH + 4 * (q ^ (u & 15))) = w1v[j];
    }
    lds_barrier();                                                                                   // #1 weight block, h1
    CSTAMP(2);

    // ================= chain waves: branch layer, head, loss, head backward, g2 -- in registers
    float dz[CAMAX] = {0.f, 0.f, 0.f, 0.f};
    cf32x4 g2T[8], dh[8];
    if (chain) {
        cf32x4 h2T[8];
        // A[m = unit 16 t + cl][k = 16 s + 4 g + {0..3}]: chunk (4 s + g) of the unit's row sits at slot (4 s + g) ^ cl
        // = 4 (s ^ (cl >> 2)) + (g ^ (cl & 3)): one address register per s, the tile t in the instruction's offset.  Two tiles at a
        // time: their chains are independent (a chain's next MFMA waits for the previous one's result)
        const float* wlane = w1s + cl * CH + 4 * (g ^ (cl & 3));
        const int c2 = cl >> 2;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            const float4 b0v = *reinterpret_cast<const float4*>(&bms[16 * t + 4 * g]), b1v = *reinterpret_cast<const float4*>(&bms[16 * t + 16 + 4 * g]);
            cf32x4 acc0 = {b0v.x, b0v.y, b0v.z, b0v.w}, acc1 = {b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float* ws = wlane + 16 * (s ^ c2);
                const float4 a0 = *reinterpret_cast<const float4*>(ws + 16 * t * CH), a1 = *reinterpret_cast<const float4*>(ws + (16 * t + 16) * CH);
                CMFMA(a0.x, h1T[s][0], acc0); CMFMA(a1.x, h1T[s][0], acc1);
                CMFMA(a0.y, h1T[s][1], acc0); CMFMA(a1.y, h1T[s][1], acc1);
                CMFMA(a0.z, h1T[s][2], acc0); CMFMA(a1.z, h1T[s][2], acc1);
                CMFMA(a0.w, h1T[s][3], acc0); CMFMA(a1.w, h1T[s][3], acc1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { h2T[t][i] = act_apply_c<ACT>(acc0[i]); h2T[t + 1][i] = act_apply_c<ACT>(acc1[i]); }
        }
        // head: z[j][row] = sum_units wh[j][unit] h2[row][unit]: A[m = j = cl][k = unit] (rows >= nout are zero), two accumulators
        cf32x4 za = {0.f, 0.f, 0.f, 0.f}, zb = za;
        const float wm = cl < CAMAX ? 1.f : 0.f;
        const float* whrow = whs + min(cl, CAMAX - 1) * CH + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            const float4 a0 = *reinterpret_cast<const float4*>(whrow + 16 * t), a1 = *reinterpret_cast<const float4*>(whrow + 16 * t + 16);
            CMFMA(a0.x * wm, h2T[t][0], za); CMFMA(a1.x * wm, h2T[t + 1][0], zb);
            CMFMA(a0.y * wm, h2T[t][1], za); CMFMA(a1.y * wm, h2T[t + 1][1], zb);
            CMFMA(a0.z * wm, h2T[t][2], za); CMFMA(a1.z * wm, h2T[t + 1][2], zb);
            CMFMA(a0.w * wm, h2T[t][3], za); CMFMA(a1.w * wm, h2T[t + 1][3], zb);
        }
        // lanes g == 0 hold z[i][row cl] in component i: hand every lane of the row's four groups the values
        float z[CAMAX];
#pragma unroll
        for (int i = 0; i < CAMAX; ++i) z[i] = __shfl(za[i] + zb[i], cl, 64) + bhs[i];

        // ---- this role's loss terms and dLoss/dz (ppo_trunk.hip, categorical branch; every group of the row computes the same numbers)
        const int m_row = m0 + row;
        const bool row_ok = m_row < M;
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        const float invM = 1.f / (float)M;
        const float4 sc = *reinterpret_cast<const float4*>(rsc + row * 4);                           // act | ret | adv | old_logp
        if (actor) {
            float adv = sc.z;
            const float old_lp = sc.w;
            asm volatile("" : "+v"(st_std));
            if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);                   // memory_tools.py:281-282
            const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
            if (row_ok) {
                const int act = (int)sc.x;
                float mx = z[0];
#pragma unroll
                for (int j = 1; j < CAMAX; ++j) if (j < A) mx = fmaxf(mx, z[j]);
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < CAMAX; ++j) if (j < A) se += expf(z[j] - mx);
                const float lse = mx + logf(se);
                float zact = z[0];
#pragma unroll
                for (int j = 1; j < CAMAX; ++j) if (j == act) zact = z[j];
                const float logp = zact - lse;
                float ent = 0.f;
#pragma unroll
                for (int j = 0; j < CAMAX; ++j) if (j < A) { const float l = z[j] - lse; ent -= expf(l) * l; }
                const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                const float ce = p.ent_coef * invM;
#pragma unroll
                for (int j = 0; j < CAMAX; ++j)
                    if (j < A) { const float l = z[j] - lse, pj = expf(l); dz[j] = s.dlogp * ((j == act ? 1.f : 0.f) - pj) + ce * pj * (l + ent); }
                t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                if (p.diag && g == 0) {
                    const int m = m_row;
                    p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                }
            }
        } else if (row_ok) {
            const float v = z[0], dv = v - sc.y;
            dz[0] = p.vf_coef * 2.f * dv * invM;
            t_c = (double)dv * dv; t_v = v;
        }
        if (g == 0) { rowstat[0 * CPT + row] = t_s; rowstat[1 * CPT + row] = t_c; rowstat[2 * CPT + row] = t_e; rowstat[3 * CPT + row] = t_v; rowstat[4 * CPT + row] = t_n; }

        // ---- g2 = (dz . W_h) * act'(h2); partial sums over this wave's 16 rows of the head-weight, head-bias and branch-bias gradients
        float* pw = partw + wave * CAMAX * CH;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            cf32x4 gg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < CAMAX; ++j) {
                if (j < nout) {
                    const float4 w = *reinterpret_cast<const float4*>(whs + j * CH + 16 * t + 4 * g);
                    gg[0] += dz[j] * w.x; gg[1] += dz[j] * w.y; gg[2] += dz[j] * w.z; gg[3] += dz[j] * w.w;
                    float4 hw;
                    hw.x = crow16_sum(dz[j] * h2T[t][0]); hw.y = crow16_sum(dz[j] * h2T[t][1]);
                    hw.z = crow16_sum(dz[j] * h2T[t][2]); hw.w = crow16_sum(dz[j] * h2T[t][3]);
                    if (cl == 0) *reinterpret_cast<float4*>(pw + j * CH + 16 * t + 4 * g) = hw;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) gg[i] *= act_grad_c<ACT>(h2T[t][i]);
            g2T[t] = gg;
            *reinterpret_cast<float4*>(g2 + row * CLD + 16 * t + 4 * g) = make_float4(gg[0], gg[1], gg[2], gg[3]);
            float4 bs;
            bs.x = crow16_sum(gg[0]); bs.y = crow16_sum(gg[1]); bs.z = crow16_sum(gg[2]); bs.w = crow16_sum(gg[3]);
            if (cl == 0) *reinterpret_cast<float4*>(partb + wave * CH + 16 * t + 4 * g) = bs;
        }
#pragma unroll
        for (int j = 0; j < CAMAX; ++j) {
            const float sb = crow16_sum(dz[j]);
            if (lane == 0) parthb[wave * 4 + j] = sb;
        }
    }
    lds_barrier();                                                                                   // #2 h1, g2, partials, rowstat
    CSTAMP(3);

    // ================= concurrently on the matrix pipe: waves 4..7 dW1 (+ the small gradients' final sums), chain waves dH1
    if (!chain) {
        const int hw = wave - 4, ht = tid - 4 * 64;                      // helper wave / helper thread 0..255
        // ---- loss terms of this (tile, role): one statistic per wave (+ the fifth on the first)
        {
            const double a = wave_sum(rowstat[hw * CPT + lane]);
            if (lane == 0) p.partials[(size_t)blockIdx.x * 8 + hw] = a;
            if (hw == 0) {
                const double b = wave_sum(rowstat[4 * CPT + lane]);
                if (lane == 0) { double* q = p.partials + (size_t)blockIdx.x * 8; q[4] = b; q[5] = 0; q[6] = 0; q[7] = 0; }
            }
        }
        // ---- head weight / bias, branch bias gradients: the four chain waves' partials in wave order
        for (int e = ht; e < nout * CH; e += 4 * 64) {
            const float v = ((partw[e] + partw[CAMAX * CH + e]) + partw[2 * CAMAX * CH + e]) + partw[3 * CAMAX * CH + e];
            slab[Lh.w_off + e] = v;
        }
        if (ht < CH) slab[L1.b_off + cb + ht] = ((partb[ht] + partb[CH + ht]) + partb[2 * CH + ht]) + partb[3 * CH + ht];
        else if (ht < CH + nout) { const int j = ht - CH; slab[Lh.b_off + j] = ((parthb[j] + parthb[4 + j]) + parthb[8 + j]) + parthb[12 + j]; }
        // ---- dW1[n][k] = sum over the 64 rows of g2[row][n] * h1[row][k]: helper wave hw owns n-tile hw, all four k-tiles
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* arow = g2 + lh * CLD + hw * 32 + li;               // A[i = n][k = row]
        const float* brow = h1 + lh * CLD + li;                         // B[k = row][j]
#pragma unroll 4
        for (int s = 0; s < CPT / 2; ++s) {
            const float av = arow[2 * s * CLD];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[2 * s * CLD + t * 32], acc[t], 0, 0, 0);
        }
        float* dW = slab + L1.w_off + (size_t)(cb + hw * 32) * CH;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int r = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)r * CH + t * 32 + li] = acc[t][rr];
            }
    } else {
        // ---- dH1^T[k][row] = sum_units W1[unit][k] g2^T[unit][row] (this role's 128 units): A[m = k = 16 s + cl][unit 16 t + 4 g + i]
        //      sits at unit * 128 + 4 ((4 s + (cl >> 2)) ^ (4 g + i)) + (cl & 3) = unit * 128 + 16 (s ^ g) + 4 ((cl >> 2) ^ i) + (cl & 3):
        //      four address registers (one per i), s and t in the instruction's offset / one XOR per s.  NOTHING but MFMAs and LDS reads in
        //      this loop: the waves beside these are issuing MFMAs too, and a vector instruction gets about one slot per MFMA there --
        //      the activation's derivative waits for the barrier.  Two k-tiles at a time (independent chains).
        const int c2 = cl >> 2, kr = cl & 3;
        const float* wi0 = w1s + (4 * g + 0) * CH + 4 * (c2 ^ 0) + kr;
        const float* wi1 = w1s + (4 * g + 1) * CH + 4 * (c2 ^ 1) + kr;
        const float* wi2 = w1s + (4 * g + 2) * CH + 4 * (c2 ^ 2) + kr;
        const float* wi3 = w1s + (4 * g + 3) * CH + 4 * (c2 ^ 3) + kr;
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            cf32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            const int o0 = 16 * (s ^ g), o1 = 16 * ((s + 1) ^ g);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int ut = 16 * t * CH;
                CMFMA(wi0[ut + o0], g2T[t][0], a0); CMFMA(wi0[ut + o1], g2T[t][0], a1);
                CMFMA(wi1[ut + o0], g2T[t][1], a0); CMFMA(wi1[ut + o1], g2T[t][1], a1);
                CMFMA(wi2[ut + o0], g2T[t][2], a0); CMFMA(wi2[ut + o1], g2T[t][2], a1);
                CMFMA(wi3[ut + o0], g2T[t][3], a0); CMFMA(wi3[ut + o1], g2T[t][3], a1);
            }
            dh[s] = a0; dh[s + 1] = a1;
        }
    }
    lds_barrier();                                                                                   // #3 nobody reads h1 / g2 any more
    CSTAMP(4);
    if (chain) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float4 o;
            o.x = dh[s][0] * act_grad_c<ACT>(h1T[s][0]); o.y = dh[s][1] * act_grad_c<ACT>(h1T[s][1]);
            o.z = dh[s][2] * act_grad_c<ACT>(h1T[s][2]); o.w = dh[s][3] * act_grad_c<ACT>(h1T[s][3]);
            *reinterpret_cast<float4*>(xb + row * CLD + 16 * s + 4 * g) = o;
        }
    }
    lds_barrier();                                                                                   // #4 g1 (this role's part)
    CSTAMP(5);
    // ================= first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c] (ppo_trunk.hip's phase: the actor's part into the
    //                   slab's first-layer region, the critic's into the fold region behind the parameters)
    {
        const int cblk = wave & 3;
        float* dst = actor ? slab : slab + p.l0_fold_off;
        const int w_at = actor ? L0.w_off : 0, b_at = actor ? L0.b_off : CH * D;
        const int kh = ((D + 3) / 4) * 2;                              // components per range, even
        const int c = cblk * 32 + li, k0 = (wave >> 2) * kh, nk = min(kh, D - k0), nq2 = (max(nk, 0) + 1) / 2;
        const float* gcol = xb + (lh * (CPT / 2)) * CLD + c;
        const float* xin = xs + (lh * (CPT / 2)) * CXLD + k0;
        float acc[CDMAX / 2], ab = 0.f;
#pragma unroll
        for (int k = 0; k < CDMAX / 2; ++k) acc[k] = 0.f;
        switch (nq2) {
            case 1: cdw0_rows<1>(gcol, xin, acc, ab); break;
            case 2: cdw0_rows<2>(gcol, xin, acc, ab); break;
            case 3: cdw0_rows<3>(gcol, xin, acc, ab); break;
            case 4: cdw0_rows<4>(gcol, xin, acc, ab); break;
            case 5: cdw0_rows<5>(gcol, xin, acc, ab); break;
            case 6: cdw0_rows<6>(gcol, xin, acc, ab); break;
            default: cdw0_rows<0>(gcol, xin, acc, ab); break;
        }
        ab += __shfl_xor(ab, 32, 64);
#pragma unroll
        for (int k = 0; k < CDMAX / 2; ++k) {
            if (k < nk) {
                const float v = acc[k] + __shfl_xor(acc[k], 32, 64);
                if (lh == 0) dst[w_at + c * D + k0 + k] = v;
            }
        }
        if (lh == 0 && (wave >> 2) == 0) dst[b_at + c] = ab;
    }
    CSTAMP(6);
#undef CSTAMP
}

// eligibility on top of ppo_trunk_eligible (ppo_trunk.hip checks the family): categorical head, A <= 4
bool ppo_chain_eligible(const xrl_ppo_fused_t& p) { return p.dist == 0 && p.A >= 1 && p.A <= CAMAX && p.pad0 == 66; }

template <int ACT>
static int launch_chain_act(const xrl_ppo_fused_t& p, hipStream_t stream) {
    const int n_tiles = (p.M + CPT - 1) / CPT;
    if (p.D == 4 && p.A == 2) hipLaunchKernelGGL((ppo_chain_kernel<ACT, 4, 2>), dim3(2 * n_tiles), dim3(FUSED_THREADS), ChainLds::BYTES, stream, p);
    else hipLaunchKernelGGL((ppo_chain_kernel<ACT, 0, 0>), dim3(2 * n_tiles), dim3(FUSED_THREADS), ChainLds::BYTES, stream, p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

int launch_ppo_chain(const xrl_ppo_fused_t& p, hipStream_t stream) {
    switch (p.layers[0].act) {
        case XRL_ACT_RELU: return launch_chain_act<XRL_ACT_RELU>(p, stream);
        case XRL_ACT_LEAKY_RELU: return launch_chain_act<XRL_ACT_LEAKY_RELU>(p, stream);
        default: return launch_chain_act<XRL_ACT_TANH>(p, stream);
    }
}

template <int ACT>
static int init_chain_act() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_chain_kernel<ACT, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ChainLds::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_chain_kernel<ACT, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, ChainLds::BYTES));
    return XRL_OK;
}

int init_ppo_chain() {
    if (int rc = init_chain_act<XRL_ACT_RELU>()) return rc;
    if (int rc = init_chain_act<XRL_ACT_LEAKY_RELU>()) return rc;
    return init_chain_act<XRL_ACT_TANH>();
}

}  // namespace xrl

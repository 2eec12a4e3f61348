// One-launch PPO minibatch for the reference's shared-trunk actor-critic family  D -> 128 -> {128 -> A | 128 -> 1}:
// Basic_MLP [128] + actor_hidden_size [128] + critic_hidden_size [128] -- every configs/ppo/classic_control/*.yaml and the MLP
// ones of configs/ppo/box2d/ (CartPole 4/2, Acrobot 6/3, MountainCar 2/3, LunarLander 8/4: Categorical_AC; Pendulum 3/1,
// BipedalWalker 24/4: Gaussian_AC with tanh on the mean) -- for D <= 24, A <= 8, categorical or Gaussian head.
//
// template <ACT, HEAD, PT>: hidden activation; HEAD 0 = categorical, 1 = Gaussian (mean as is), 2 = Gaussian (tanh on the mean);
// PT = rows per workgroup.  Workgroup = (PT-row tile, role): role 0 owns the actor branch (rows 0..127 of the stacked branch
// layer, the A head rows, log_std), role 1 the critic branch -- they need nothing from each other (csrc/ppo_split.hip, round 2):
// the shared first layer is recomputed by both, every other parameter belongs to one role (disjoint slab regions), and the
// first-layer gradient, linear in dLoss/dh1 = actor part + critic part, is formed per role: the actor's in the slab's
// first-layer region, the critic's in a fold region behind the parameters that the reduction adds onto it.
//   PT = 32 (16 threads per row in the VALU phases, two workgroups per CU): small minibatches -- the yaml defaults give 320 rows;
//   PT = 64 (8 threads per row, two 32-row MFMA blocks per workgroup, one workgroup per CU): half the weight stream per CU and
//            half the gradient slabs per row -- minibatches of >= 128 32-row tiles (round 3; the CartPole headline).
// This file replaces round 2's ppo_split.hip and round 3's ppo_pair.hip (its (4, 2, categorical) instances ARE those kernels:
// same MFMA chains, same reduction trees per element).  The small parameters (first layer, biases, heads, log_std: <= 20 KB)
// come straight from the flat parameter buffer into LDS, so the class needs no packed image and no mirror map of its own; only
// the 128 -> 256 branch layer is read in MFMA B-fragment order (xrl_pack_mid_frags copy, kept current by the optimiser launch).
// Rows: gathered per launch through idx from the rollout buffer's fields, or -- D = 4, categorical -- from the 32-byte records
// of xrl_pack_transitions / xrl_gather_rows.
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward),
// distributions.py:128-192 (Categorical / DiagGaussian log_prob, entropy), actor_head.py:14-72.
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"
#include "opt_chain.h"

namespace xrl {

// second kernel argument: the optimiser step of the previous minibatch (CHAIN instances, xrl_ppo_trunk_chained) or nothing
template <bool CHAIN> struct trunk_chain_arg { typedef xrl_opt_chain_t type; };
struct trunk_no_chain { int unused; };
template <> struct trunk_chain_arg<false> { typedef trunk_no_chain type; };

typedef unsigned tu32x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 128;                      // hidden width (trunk and each branch)
constexpr int TLD = TH + 4;                  // row stride of every LDS level
constexpr int TDMAX = 24, TXLD = 28;         // observation width limit / row stride of the gathered observations
constexpr int TAMAX = 8;                     // head width limit

template <int PT>
struct TrunkLds {
    // h1, h2 (the first-layer gradient's partial sums take h2's place once the head gradients have been formed), g2; gathered rows; parameters
    static constexpr int H1 = 0, H2 = H1 + PT * TLD, G2 = H2 + PT * TLD, XS = G2 + PT * TLD, RSC = XS + PT * TXLD,
                         DZH = RSC + PT * 12, W0T = DZH + PT * 16, B0 = W0T + TDMAX * TLD, BM = B0 + TH, WH = BM + TH,
                         BH = WH + TAMAX * TLD, LS = BH + TAMAX, SRC = LS + TAMAX, FLOATS = SRC + PT;
    static constexpr int BYTES = FLOATS * 4 + PT * 5 * 8;
};

template <int TPR>
__device__ __forceinline__ float trow_sum(float v) {                   // sum over the TPR (8 | 16) consecutive lanes of a row
    if (TPR == 16) v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}

// DS / AS: compile-time observation / head width of an instance (0 = taken from the arguments): the CartPole class (4, 2) -- the
// headline -- gets its loops resolved at compile time.  CHAIN: the launch's workgroups first finish the optimiser step of the
// minibatch before this one (csrc/opt_chain.h) and run on the parameters it leaves.
template <int ACT, int HEAD, int PT, int DS, int AS, bool CHAIN>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_trunk_kernel(xrl_ppo_fused_t p, typename trunk_chain_arg<CHAIN>::type o) {
    using L = TrunkLds<PT>;
    constexpr int TPR = FUSED_THREADS / PT;            // threads per row in the VALU phases: 16 | 8
    constexpr int CPT = TH / TPR;                      // first-layer columns per thread: 8 | 16
    constexpr int NCH = (TH / 4) / TPR;                // float4 chunks of the branch level per thread: 2 | 4
    constexpr int RB = PT / 32;                        // 32-row MFMA blocks per workgroup: 1 | 2
    constexpr bool GAUSS = HEAD != 0;
    constexpr int OACT = HEAD == 2 ? XRL_ACT_TANH : XRL_ACT_NONE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds + L::H1;
    float* h2 = lds + L::H2;
    float* g2 = lds + L::G2;
    float* xs = lds + L::XS;                           // [PT][28] gathered observations, zero beyond D
    float* rsc = lds + L::RSC;                         // [PT][12] act[<= 8] | ret | adv | old_logp
    float* dzh = lds + L::DZH;                         // [PT][16] dLoss/d(head pre-activations)[8] | d log_std terms[8]
    float* w0t = lds + L::W0T;                         // [D][132] first-layer weights, k-major
    float* b0s = lds + L::B0;
    float* bms = lds + L::BM;                          // this role's branch bias
    float* whs = lds + L::WH;                          // [nout][132] this role's head rows
    float* bhs = lds + L::BH;
    float* lss = lds + L::LS;                          // log_std
    double* rowstat = reinterpret_cast<double*>(lds + L::FLOATS);       // [5][PT] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_fused_t) + (CHAIN ? sizeof(xrl_opt_chain_t) : 0)>();
    const int tid = threadIdx.x, M = p.M, D = DS ? DS : p.D, A = AS ? AS : p.A;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cblk = wave & 3, rblk = RB == 2 ? (wave >> 2) : 0;       // MFMA phases: 32-column block / 32-row block of this wave
    const int tile = blockIdx.x >> 1, role = blockIdx.x & 1;
    const bool actor = role == 0;
    const int nout = actor ? A : 1;
    const int cb = role * TH;                          // this role's first column of the stacked branch level
    const int m0 = tile * PT;
    const int r = tid / TPR, sub = tid % TPR, m_row = m0 + r;
    const bool row_ok = m_row < M;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
    const xrl_fused_layer_t& Lh = actor ? La : Lc;

    long long* dbg = p.dbg;                            // diagnostics (tools/probe_pair_phases.py), see the end of the kernel
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - 1;
#define TSTAMP(k) do { if (dbg_me) dbg[k] = clock64(); } while (0)
    // per-wave stamps of the last workgroup (dbg[1100 + 16 wave + k]): where each wave is inside the backward phases
    const bool dbg_w = dbg && (tid & 63) == 0 && blockIdx.x == gridDim.x - 1;
#define WSTAMP(k) do { if (dbg_w) dbg[1100 + 16 * (tid >> 6) + (k)] = clock64(); } while (0)
    if (dbg && tid == 0) dbg[16 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    TSTAMP(0);

    // ================= loads: this role's 64 KB of W1 B-fragments (waves w and w + 4 of a 64-row tile fetch the same lines), the
    //                   small parameters, the rows
    float4 pf[PD];
    const bool records = !GAUSS && D == 4 && (p.f_rows || p.f_packed);
    if (records) {                                     // 32-byte records obs[4] | act | ret | adv | old_logp: one wave
        if (wave == 7 && lane < PT) {
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
            *reinterpret_cast<float4*>(xs + lane * TXLD) = xr;
            *reinterpret_cast<float4*>(xs + lane * TXLD + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            rsc[lane * 12 + 0] = sc.x; rsc[lane * 12 + 8] = sc.y; rsc[lane * 12 + 9] = sc.z; rsc[lane * 12 + 10] = sc.w;
        }
    }
    // rows through idx from the buffer's FIELDS: thread (row r, sub) takes the observation elements sub, sub + TPR, ... of its row and the
    // scalar slots sub, sub + TPR, ... (slots 0..7 action components, 8 return, 9 advantage, 10 old log-prob) -- every thread resolves its
    // row's buffer index itself (env-major flat index -> time-major row, memory_tools.py:270) and ALL gathers of the tile are in flight
    // together (round 6: as loops over the tile behind a shared index array they were three dependent round trips to scattered rows)
    constexpr int GXQ = (TXLD + TPR - 1) / TPR, GSQ = (12 + TPR - 1) / TPR;
    float g_x[GXQ], g_s[GSQ];
#pragma unroll
    for (int qx = 0; qx < GXQ; ++qx) g_x[qx] = 0.f;
#pragma unroll
    for (int qs = 0; qs < GSQ; ++qs) g_s[qs] = 0.f;
    if (!records) {
        int src = -1;
        if (row_ok) {
            const int64_t fl = p.idx[m_row];
            const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
            src = t * p.n_envs + env;
        }
        if (src >= 0) {
            const int na = GAUSS ? A : 1;
#pragma unroll
            for (int qx = 0; qx < GXQ; ++qx) {
                const int k = sub + qx * TPR;
                if (k < D) g_x[qx] = p.f_obs[(size_t)src * D + k];
            }
#pragma unroll
            for (int qs = 0; qs < GSQ; ++qs) {
                const int k = sub + qs * TPR;
                if (k < na) g_s[qs] = p.f_act[(size_t)src * na + k];
                else if (k == 8) g_s[qs] = p.f_ret[src];
                else if (k == 9) g_s[qs] = p.f_adv[src];
                else if (k == 10) g_s[qs] = p.f_logp[src];
            }
        }
    }
    // (the rows above depend on no parameter: requested before the optimiser step of the previous minibatch, which every load
    //  below has to wait for)
    if constexpr (CHAIN) chain_prologue(o, h1, dbg ? dbg + 2048 : nullptr);      // (diagnostics: dbg holds >= 2048 + 8 x workgroups words then)
    float st_mean = 0.f, st_std = 1.f;
    if (p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    // small parameters straight from the flat buffer: W0 [128][D] -> k-major, biases, this role's head rows, log_std.  EVERY request
    // first (a fixed number per thread), the LDS writes behind the fragment stream's requests: written as loops over the arrays the
    // compiler emitted load -> s_waitcnt vmcnt(0) -> ds_write per array -- five L2 round trips in a row in front of the stream
    // (round 6, found in csrc/ppo_trunk_bx.hip's listing; same values, same LDS image)
    constexpr int W0Q = DS ? (TH * DS + FUSED_THREADS - 1) / FUSED_THREADS : (TH * TDMAX + FUSED_THREADS - 1) / FUSED_THREADS;
    constexpr int WHQ = AS ? (TH * AS + FUSED_THREADS - 1) / FUSED_THREADS : (TH * TAMAX + FUSED_THREADS - 1) / FUSED_THREADS;
    float w0v[W0Q], whv[WHQ], smv = 0.f, lsv = 0.f;
#pragma unroll
    for (int q = 0; q < W0Q; ++q) {
        const int e = tid + q * FUSED_THREADS;
        w0v[q] = e < TH * D ? p.params[L0.w_off + e] : 0.f;
    }
    if (tid < TH) smv = p.params[L0.b_off + tid];
    else if (tid < 2 * TH) smv = p.params[L1.b_off + cb + tid - TH];
    else if (tid < 2 * TH + TAMAX) {
        const int j = tid - 2 * TH;
        if (j < nout) smv = p.params[Lh.b_off + j];
        if (GAUSS && j < A) lsv = p.params[p.log_std_off + j];
    }
#pragma unroll
    for (int q = 0; q < WHQ; ++q) {
        const int e = tid + q * FUSED_THREADS;
        whv[q] = e < nout * TH ? p.params[Lh.w_off + e] : 0.f;
    }
    // (the fragment stream is requested AFTER the small loads: a wave's loads retire in order, and the first layer must not wait
    //  for 64 KB of weights it does not read)
    {
        const int t = 4 * role + cblk;                                   // tile of the stacked 256-row W1
        const float* base = p.frag_image + ((size_t)t * (TH / 8) * 64 + lane) * 4;
#pragma unroll
        for (int q = 0; q < PD; ++q) pf[q] = *reinterpret_cast<const float4*>(base + frag_slot(q, t, TH / 8, 1) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < W0Q; ++q) {
        const int e = tid + q * FUSED_THREADS;
        if (e < TH * D) { const int c = e / D, k = e - c * D; w0t[k * TLD + c] = w0v[q]; }
    }
    if ((D & 1) && tid < TH) w0t[D * TLD + tid] = 0.f;    // (the first layer's MFMAs walk k in pairs)
    if (tid < TH) b0s[tid] = smv;
    else if (tid < 2 * TH) bms[tid - TH] = smv;
    else if (tid < 2 * TH + TAMAX) { bhs[tid - 2 * TH] = smv; lss[tid - 2 * TH] = lsv; }
#pragma unroll
    for (int q = 0; q < WHQ; ++q) {
        const int e = tid + q * FUSED_THREADS;
        if (e < nout * TH) whs[(e >> 7) * TLD + (e & (TH - 1))] = whv[q];
    }
    if (!records) {
        // observations (zero padded to 28 columns), actions, ret | adv | old_logp
#pragma unroll
        for (int qx = 0; qx < GXQ; ++qx) {
            const int k = sub + qx * TPR;
            if (k < TXLD) xs[r * TXLD + k] = g_x[qx];
        }
#pragma unroll
        for (int qs = 0; qs < GSQ; ++qs) {
            const int k = sub + qs * TPR;
            if (k < 12) rsc[r * 12 + k] = g_s[qs];
        }
    }
    lds_barrier();                                                                                   // #0 rows, parameters
    TSTAMP(1);

    // ================= forward: first layer on the matrix cores (round 4; the VALU form -- a k-ordered fma chain per element,
    //                   the same numbers -- took 3.7 k cycles of LDS reads at 64 rows): wave (cblk, rblk), ceil(D / 2) MFMAs
    if (RB == 2 || wave < 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* xr = xs + (rblk * 32 + li) * TXLD + lh;            // A[i = row][k = lh + 2 s] (zero beyond D)
        const float* wk = w0t + lh * TLD + cblk * 32 + li;              // B[k = lh + 2 s][j = column]
        const int ns = (D + 1) >> 1;
        if (DS == 4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0], wk[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[2], wk[2 * TLD], acc, 0, 0, 0);
        } else {
            for (int s2 = 0; s2 < ns; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[2 * s2], wk[2 * s2 * TLD], acc, 0, 0, 0);
        }
        const int col = cblk * 32 + li;
        const float b0v = b0s[col];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h1[row * TLD + col] = act_apply_c<ACT>(acc[rr] + b0v);
        }
    }
    lds_barrier();                                                                                   // #1 h1
    TSTAMP(2);
    // ---- this role's branch layer 128 -> 128 on the matrix cores: wave (cblk, rblk) owns columns [32 cblk, +32) of rows [32 rblk, +32)
    //      (PT = 32: waves 0..3 only, as in ppo_split_kernel)
    if (RB == 2 || wave < 4) {
        const float* arow = h1 + (rblk * 32 + li) * TLD + 4 * lh;
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
        const float bm = bms[col];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * TLD + col] = act_apply_c<ACT>(acc[rr] + bm);
        }
        // forward fragments consumed: the same registers take the BACKWARD section of the fragment copy for dH1 below (output tile
        // kt = cblk, this role's 16 n-chunks q = 16 role + i; xrl_pack_mid_frags) -- a second stream that has the head / loss /
        // weight-gradient phases to arrive, instead of transposing the forward fragments through LDS (measured: csrc/ppo_wide.hip)
        {
            const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_image), 0, 2 * 2 * TH * TH * 4, 0x00020000);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                const int q = 16 * role + i;
                const tu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (2 * TH * TH + (cblk * 32 + frag_slot(q, cblk, 32, 2)) * 256) * 4, 0);
                pf[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                                                                   // #2 h2
    TSTAMP(3);

    // ================= head forward (VALU, TPR threads per row), this role's loss terms, head backward -- in registers.
    // k-chunks q = sub + TPR i (the 32 float4 chunks of this role's 128 columns).  After the all-reduce every thread of a row holds
    // the row's head pre-activations; thread `sub` owns action dimension `sub` of the Gaussian loss (as in ppo_wide_kernel).
    float4 a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = *reinterpret_cast<const float4*>(h2 + r * TLD + 4 * (sub + TPR * i));
    float z[TAMAX];
#pragma unroll
    for (int j = 0; j < TAMAX; ++j) {
        z[j] = 0.f;
        if (j < nout) {
            float c = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(whs + j * TLD + 4 * (sub + TPR * i));
                c += a[i].x * w.x + a[i].y * w.y + a[i].z * w.z + a[i].w * w.w;
            }
            z[j] = trow_sum<TPR>(c) + bhs[j];
        }
    }
    {
        float dz[TAMAX], my_gls = 0.f;
#pragma unroll
        for (int j = 0; j < TAMAX; ++j) dz[j] = 0.f;
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        const float invM = 1.f / (float)M;
        if (actor) {
            float adv = rsc[r * 12 + 9];
            const float old_lp = rsc[r * 12 + 10];
            asm volatile("" : "+v"(st_std));
            if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);                   // memory_tools.py:281-282
            const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
            if (!GAUSS) {
                if (row_ok) {
                    const int act = (int)rsc[r * 12];
                    float mx = z[0];
#pragma unroll
                    for (int j = 1; j < TAMAX; ++j) if (j < A) mx = fmaxf(mx, z[j]);
                    float se = 0.f;
#pragma unroll
                    for (int j = 0; j < TAMAX; ++j) if (j < A) se += expf(z[j] - mx);
                    const float lse = mx + logf(se);
                    float zact = z[0];
#pragma unroll
                    for (int j = 1; j < TAMAX; ++j) if (j == act) zact = z[j];
                    const float logp = zact - lse;
                    float ent = 0.f;
#pragma unroll
                    for (int j = 0; j < TAMAX; ++j) if (j < A) { const float l = z[j] - lse; ent -= expf(l) * l; }
                    const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                    const float ce = p.ent_coef * invM;
#pragma unroll
                    for (int j = 0; j < TAMAX; ++j)
                        if (j < A) { const float l = z[j] - lse, pj = expf(l); dz[j] = s.dlogp * ((j == act ? 1.f : 0.f) - pj) + ce * pj * (l + ent); }
                    t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                    if (p.diag && sub == 0) {
                        const int m = m_row;
                        p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                    }
                }
            } else {
                // DiagGaussianDistribution (distributions.py:155-192): log_prob / entropy summed over the action dims; dim `sub`
                const bool mine = sub < A;
                float zmine = 0.f;
#pragma unroll
                for (int j = 0; j < TAMAX; ++j) if (j == sub) zmine = z[j];
                float mu = 0.f, df = 0.f, var = 1.f, term = 0.f, entj = 0.f;
                if (mine) {
                    mu = act_apply_c<OACT>(zmine);                                                   // activation_action (actor_head.py:62)
                    const float ls = lss[sub], sd = expf(ls);
                    var = sd * sd; df = rsc[r * 12 + sub] - mu;
                    term = -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
                    entj = 0.5f + LOG_SQRT_2PI + logf(sd);
                }
                const float logp = trow_sum<TPR>(term), ent = trow_sum<TPR>(entj);
                float my_dz = 0.f;
                if (row_ok) {
                    const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                    if (mine) {
                        my_dz = (s.dlogp * df / var) * act_grad_c<OACT>(mu);
                        my_gls = s.dlogp * (df * df / var - 1.f);
                    }
                    t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                    if (p.diag && sub == 0) {
                        const int m = m_row;
                        p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                    }
                }
                if (sub < TAMAX) { dzh[r * 16 + sub] = my_dz; dzh[r * 16 + 8 + sub] = my_gls; }
                // the row's threads sit in one wave and a wave's LDS operations execute in order: the values are there
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float4 d0 = *reinterpret_cast<const float4*>(dzh + r * 16), d1 = *reinterpret_cast<const float4*>(dzh + r * 16 + 4);
                dz[0] = d0.x; dz[1] = d0.y; dz[2] = d0.z; dz[3] = d0.w; dz[4] = d1.x; dz[5] = d1.y; dz[6] = d1.z; dz[7] = d1.w;
            }
        } else if (row_ok) {
            const float v = z[0], dv = v - rsc[r * 12 + 8];
            dz[0] = p.vf_coef * 2.f * dv * invM;
            t_c = (double)dv * dv; t_v = v;
        }
        if (sub == 0) {
            if (!(actor && GAUSS)) {
#pragma unroll
                for (int j = 0; j < TAMAX; ++j) { dzh[r * 16 + j] = dz[j]; dzh[r * 16 + 8 + j] = 0.f; }
            }
            rowstat[0 * PT + r] = t_s; rowstat[1 * PT + r] = t_c; rowstat[2 * PT + r] = t_e; rowstat[3 * PT + r] = t_v; rowstat[4 * PT + r] = t_n;
        }
        // dH2 = dZh . W_h, times act'(h2): this thread's k-chunks
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < TAMAX; ++j) {
                if (j < nout) {
                    const float4 w = *reinterpret_cast<const float4*>(whs + j * TLD + 4 * (sub + TPR * i));
                    g.x += dz[j] * w.x; g.y += dz[j] * w.y; g.z += dz[j] * w.z; g.w += dz[j] * w.w;
                }
            }
            g.x *= act_grad_c<ACT>(a[i].x); g.y *= act_grad_c<ACT>(a[i].y); g.z *= act_grad_c<ACT>(a[i].z); g.w *= act_grad_c<ACT>(a[i].w);
            *reinterpret_cast<float4*>(g2 + r * TLD + 4 * (sub + TPR * i)) = g;
        }
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    TSTAMP(4);

    // ================= backward
    // ---- loss terms of this (tile, role): one lane per row; the actor fills surrogate / entropy / clip count, the critic the value
    //      terms -- the partials reduction adds the rows of all workgroups
    // (one statistic per wave: next to another wave's MFMAs a wave gets a vector issue slot per MFMA -- five reductions in one wave
    //  took ~10 k cycles there and held the workgroup's last barrier)
    if (wave < 5) {
        double a = lane < PT ? rowstat[wave * PT + lane] : 0.0;
        a = wave_sum(a);
        if (lane == 0) p.partials[(size_t)blockIdx.x * 8 + wave] = a;
    } else if (wave == 5 && lane < 3) p.partials[(size_t)blockIdx.x * 8 + 5 + lane] = 0.0;
    // ---- head weight / bias gradients, log_std gradient, this role's branch-layer bias gradient: VALU sums over the PT rows, rows in
    //      order.  Measured in round 4 (tools/probe_pair_phases.py, per-wave stamps): next to another wave's MFMAs a wave gets
    //      about one vector issue slot per MFMA: a wave WITHOUT small gradients used to start the weight-gradient MFMAs below at once
    //      and the loops of the wave it shares a SIMD with crawled (10 k cycles instead of 2-3 k, holding the workgroup's last
    //      barrier) -- hence the barrier behind this phase.  Dealing the elements evenly to all eight waves (5.0 k cycles, every
    //      wave) and carrying them inside the MFMA loop (13 k instead of 7.5 k for the loop) were both slower than this form.
    for (int e = tid; e < nout * TH; e += FUSED_THREADS) {
        const int j = e >> 7, k = e & (TH - 1);
        const float* hp = h2 + k;
        float acc = 0.f;
#pragma unroll 16
        for (int rr = 0; rr < PT; ++rr) acc += dzh[rr * 16 + j] * hp[rr * TLD];
        slab[Lh.w_off + e] = acc;
    }
    // (bias gradients: plain column sums of per-row terms, summed in DOUBLE and rounded once per tile.  With advantages normalised to
    //  mean 0 the head bias' 8 192 terms cancel to 6e-5 of sum|terms| (tests/test_oracle_head_bias_floor.py): 64 float adds per tile
    //  left the sum 5.7e-5 of its own size from the float64 value -- as far as torch's float32 sum -- where float32 TERMS summed
    //  exactly land at < 2e-5; the slabs are summed in double by xrl_reduce_adam already)
    if (tid >= 4 * 64 && tid < 4 * 64 + TH) {
        const int t = tid - 4 * 64;
        double acc0 = 0.0;
#pragma unroll 16
        for (int rr = 0; rr < PT; ++rr) acc0 += (double)g2[rr * TLD + t];
        slab[L1.b_off + cb + t] = (float)acc0;
    } else if (tid >= 6 * 64 && tid < 6 * 64 + 2 * TAMAX) {
        const int t = tid - 6 * 64;                                      // 0..7 head bias, 8..15 log_std
        if (t < nout || (t >= 8 && GAUSS && actor && t - 8 < A)) {
            double acc = 0.0;
#pragma unroll 16
            for (int rr = 0; rr < PT; ++rr) acc += (double)dzh[rr * 16 + t];
            // d(-ent_coef * mean_m sum_j(log_std_j + c)) / d log_std_j = -ent_coef, added once (tile 0), as in ppo_loss.hip / ppo_wide.hip
            if (t >= 8 && tile == 0) acc -= (double)p.ent_coef;
            if (t < 8) slab[Lh.b_off + t] = (float)acc;
            else slab[p.log_std_off + t - 8] = (float)acc;
        }
    }
    WSTAMP(0);
    lds_barrier();                                                                                   // #3b no MFMA beside a vector loop
    TSTAMP(5);
    // ---- dW1[n][k] = sum over the PT rows of g2[row][n] * h1[row][k] for this role's 128 rows n: 4 x 4 tiles of 32 x 32, wave w owns
    //      n-tile (w & 3) and the k-tiles 2 (w >> 2), 2 (w >> 2) + 1; PT / 2 chained MFMAs per tile, rows in order
    f32x16 dacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) dacc[i] = 0.f;
    {
        const int nt = wave & 3, kt0 = 2 * (wave >> 2);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* arow = g2 + lh * TLD + nt * 32 + li;               // A[i = n][k = row]
        const float* brow = h1 + lh * TLD + kt0 * 32 + li;              // B[k = row][j]
#pragma unroll 8
        for (int s = 0; s < PT / 2; ++s) {
            const float av = arow[2 * s * TLD];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bv = brow[2 * s * TLD + t * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
        if (dbg_w) { float t_ = acc[1][15]; asm volatile("" ::"v"(t_)); }
        TSTAMP(6);
        WSTAMP(1);
        // ---- this role's part of dH1 = g2 . W1 (sum over its 128 rows n): wave (cblk, rblk) owns output columns [32 cblk, +32) of
        //      rows [32 rblk, +32); B operand = the backward fragments requested after the forward layer.  The result (times act'(h1))
        //      goes where h2 was: every wave is past its last read of h2 once it is through the barrier below.
        //      The 32 stores of this wave's dW1 tiles are issued BETWEEN these MFMAs, one per two (round 4: issued in one burst behind
        //      the weight-gradient loop, the eight waves' 64 KB queued on the CU's ~10 B/clk store path and the waves stood 1-6 k
        //      cycles in front of their next MFMA).
        float* dW = slab + L1.w_off + (size_t)(cb + nt * 32) * TH;
        if (RB == 2 || wave < 4) {
            const float* arow2 = g2 + (rblk * 32 + li) * TLD + 4 * lh;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                float4 af[PD / 2];
#pragma unroll
                for (int i = 0; i < PD / 2; ++i) af[i] = *reinterpret_cast<const float4*>(arow2 + (hq * 8 + i) * 8);
#pragma unroll
                for (int i = 0; i < PD / 2; ++i) { MFMA4(af[i], pf[hq * 8 + i], dacc) }
            }
        }
        if (!(p.pad3 & 1)) {                                             // (pad3 bit 0: diagnostics -- tools/probe_pair_sequence.py times the pair without these stores)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)row * TH + (kt0 + t) * 32 + li] = acc[t][rr];
            }
        }
        if (RB == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); }
        }
    }
    if (dbg_w) { float t_ = dacc[15]; asm volatile("" ::"v"(t_)); }
    WSTAMP(3);
    WSTAMP(4);
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c], g1 = dH1 * act'(h1) -- straight from the dH1 accumulators
    //      (round 4: g1 used to go to LDS ([row][column] where h2 was), two barriers around it, and come back through column loops: 4.4 k
    //      cycles): lane (li, lh) of wave (cblk, rblk) holds 16 rows of column 32 cblk + li; it sums them against the rows' observations
    //      (every lane of a half-wave reads the same row of xs: broadcasts), the halves meet by a lane shuffle, the two row blocks of a
    //      64-row tile through LDS (where h2 was: nobody reads it after barrier #3b).  The actor's part goes into the slab's first-layer
    //      region, the critic's into the fold region behind the parameters (the reduction adds it onto the same columns).
    {
        float* dst = actor ? slab : slab + p.l0_fold_off;
        const int w_at = actor ? L0.w_off : 0, b_at = actor ? L0.b_off : TH * D;
        constexpr int NQ = DS ? (DS + 3) / 4 : TDMAX / 4;               // float4 chunks of an observation row
        float* part = h2;                                                // [RB][128][TDMAX + 1] partial sums of the row blocks
        constexpr int PLD = TDMAX + 1;
        if (RB == 2 || wave < 4) {
            const int c = cblk * 32 + li;
            // (round 6: any-D instances walk the observation row in passes of two float4 chunks -- 8 accumulators live instead of 24:
            //  the 64-row any-(D, A) instances spilled 12-13 VGPRs here; same sums, element by element, rows in the same order)
            constexpr int NQP = DS ? NQ : 2, NPASS = NQ / NQP;
            float ab = 0.f;
            float* pp = part + (rblk * TH + c) * PLD;
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass) {
                const int q0 = pass * NQP;
                if (pass > 0 && 4 * q0 >= D) break;
                float acc[4 * NQP];
#pragma unroll
                for (int k = 0; k < 4 * NQP; ++k) acc[k] = 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                    const float g = dacc[rr] * act_grad_c<ACT>(h1[row * TLD + c]);
                    if (pass == 0) ab += g;
#pragma unroll
                    for (int q = 0; q < NQP; ++q) {
                        if (4 * (q0 + q) < D) {
                            const float4 x = *reinterpret_cast<const float4*>(xs + row * TXLD + 4 * (q0 + q));    // (zero beyond D)
                            acc[4 * q] += g * x.x; acc[4 * q + 1] += g * x.y; acc[4 * q + 2] += g * x.z; acc[4 * q + 3] += g * x.w;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4 * NQP; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
                if (lh == 0) {
#pragma unroll
                    for (int k = 0; k < 4 * NQP; ++k) {
                        if (4 * q0 + k < D) {
                            if (RB == 1) dst[w_at + c * D + 4 * q0 + k] = acc[k];
                            else pp[4 * q0 + k] = acc[k];
                        }
                    }
                }
            }
            ab += __shfl_xor(ab, 32, 64);
            if (lh == 0) {
                if (RB == 1) dst[b_at + c] = ab;
                else pp[TDMAX] = ab;
            }
        }
        if (RB == 2) {
            WSTAMP(5);
            lds_barrier();                                                                           // #4 the two row blocks' partial sums
            TSTAMP(7);
            WSTAMP(6);
            for (int e = tid; e < TH * (D + 1); e += FUSED_THREADS) {
                const int c = e / (D + 1), k = e - c * (D + 1), kk = k < D ? k : TDMAX;
                const float v = part[c * PLD + kk] + part[(TH + c) * PLD + kk];
                dst[k < D ? w_at + c * D + k : b_at + c] = v;
            }
        }
    }
    TSTAMP(8);
    WSTAMP(7);
    if (dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's stores are out
        dbg[17 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef TSTAMP
#undef WSTAMP
}

// the family: widths [D, 128, 256, A + 1] with D <= 24, A <= 8, the first layer at the front of the flat layout (the fold region maps
// onto columns [0, 128 D + 128)), the fragment copy of the branch layer present, hidden activation relu / leaky_relu / tanh
extern bool g_fast_enabled_ppo;
bool ppo_trunk_eligible(const xrl_ppo_fused_t& p) {
    if (p.n_layers != 4 || p.n_head_layers != 2 || p.n_levels != 4 || !p.frag_image || p.l0_fold_off <= 0) return false;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
    const int D = p.D, A = p.A;
    if (D < 1 || D > TDMAX || A < 1 || A > TAMAX) return false;
    if (L0.K != D || L0.N != TH || L1.K != TH || L1.N != 2 * TH || La.K != TH || La.N != A || Lc.K != TH || Lc.N != 1) return false;
    if (La.in_off != 0 || Lc.in_off != TH || L0.w_off != 0 || L0.b_off != TH * D) return false;
    if (L0.act != L1.act || (L0.act != XRL_ACT_RELU && L0.act != XRL_ACT_LEAKY_RELU && L0.act != XRL_ACT_TANH)) return false;
    if (p.dist != 0 && p.dist != 1) return false;
    if (p.dist == 1 && (p.log_std_off <= 0 || (p.out_act != XRL_ACT_NONE && p.out_act != XRL_ACT_TANH))) return false;
    if ((p.l0_fold_off & 3) || p.l0_fold_off + TH * D + TH > p.slab_stride) return false;
    return p.pad0 == 0 || p.pad0 == 32 || p.pad0 == 64;   // (66 was round 4's register-chained variant: measured slower, tools/csrc/ppo_chain.hip)
}

// (CHAIN instances exist for the CartPole class -- categorical head, (D, A) = (4, 2), the headline's -- only: the chained launch
//  measured slower than the launch pair (DESIGN.md section 3 "Round 6"), so it stays an option of that class, not of the family)
template <int ACT, int HEAD, int DS, int AS>
static int launch_trunk_pt(const xrl_ppo_fused_t& p, const xrl_opt_chain_t* o, hipStream_t stream) {
    constexpr bool HAS_CHAIN = HEAD == 0 && DS == 4 && AS == 2;
    if (o && !HAS_CHAIN) { set_error("xrl_ppo_trunk_chained: only the categorical (4, 2) class has chained instances"); return XRL_EINVAL; }
    if (p.pad0 == 64) {
        const int n_tiles = (p.M + 63) / 64;
        if constexpr (HAS_CHAIN) {
            if (o) { hipLaunchKernelGGL((ppo_trunk_kernel<ACT, HEAD, 64, DS, AS, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), TrunkLds<64>::BYTES, stream, p, *o); XRL_CHECK_LAUNCH(); return XRL_OK; }
        }
        hipLaunchKernelGGL((ppo_trunk_kernel<ACT, HEAD, 64, DS, AS, false>), dim3(2 * n_tiles), dim3(FUSED_THREADS), TrunkLds<64>::BYTES, stream, p, trunk_no_chain{0});
    } else {
        const int n_tiles = (p.M + 31) / 32;
        if constexpr (HAS_CHAIN) {
            if (o) { hipLaunchKernelGGL((ppo_trunk_kernel<ACT, HEAD, 32, DS, AS, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), TrunkLds<32>::BYTES, stream, p, *o); XRL_CHECK_LAUNCH(); return XRL_OK; }
        }
        hipLaunchKernelGGL((ppo_trunk_kernel<ACT, HEAD, 32, DS, AS, false>), dim3(2 * n_tiles), dim3(FUSED_THREADS), TrunkLds<32>::BYTES, stream, p, trunk_no_chain{0});
    }
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

template <int ACT>
static int launch_trunk_head(const xrl_ppo_fused_t& p, const xrl_opt_chain_t* o, hipStream_t stream) {
    if (p.dist == 0) return (p.D == 4 && p.A == 2) ? launch_trunk_pt<ACT, 0, 4, 2>(p, o, stream) : launch_trunk_pt<ACT, 0, 0, 0>(p, o, stream);
    return p.out_act == XRL_ACT_TANH ? launch_trunk_pt<ACT, 2, 0, 0>(p, o, stream) : launch_trunk_pt<ACT, 1, 0, 0>(p, o, stream);
}

int launch_ppo_trunk(const xrl_ppo_fused_t& p, const xrl_opt_chain_t* o, hipStream_t stream) {
    switch (p.layers[0].act) {
        case XRL_ACT_RELU: return launch_trunk_head<XRL_ACT_RELU>(p, o, stream);
        case XRL_ACT_LEAKY_RELU: return launch_trunk_head<XRL_ACT_LEAKY_RELU>(p, o, stream);
        default: return launch_trunk_head<XRL_ACT_TANH>(p, o, stream);
    }
}

template <int ACT, int HEAD, int DS, int AS>
static int init_trunk_one() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_kernel<ACT, HEAD, 32, DS, AS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TrunkLds<32>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_kernel<ACT, HEAD, 64, DS, AS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TrunkLds<64>::BYTES));
    if constexpr (HEAD == 0 && DS == 4 && AS == 2) {
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_kernel<ACT, HEAD, 32, DS, AS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TrunkLds<32>::BYTES));
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_kernel<ACT, HEAD, 64, DS, AS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TrunkLds<64>::BYTES));
    }
    return XRL_OK;
}

int init_ppo_trunk() {
#define TRUNK_INIT(a) \
    if (int rc = init_trunk_one<a, 0, 0, 0>()) return rc; \
    if (int rc = init_trunk_one<a, 0, 4, 2>()) return rc; \
    if (int rc = init_trunk_one<a, 1, 0, 0>()) return rc; \
    if (int rc = init_trunk_one<a, 2, 0, 0>()) return rc;
    TRUNK_INIT(XRL_ACT_RELU)
    TRUNK_INIT(XRL_ACT_LEAKY_RELU)
    TRUNK_INIT(XRL_ACT_TANH)
#undef TRUNK_INIT
    return XRL_OK;
}

// Workgroups of a chained launch that can be resident at once (its barriers spin): the runtime's occupancy figure for the
// headline instance (every instance of the family has the same LDS footprint per tile size; registers differ by a few) times the
// compute units, with the guide's caveat that the API may answer one block per CU high near an SGPR edge: at 134 KB (64-row tiles)
// or 77 KB (32-row tiles) of LDS per workgroup LDS is the limit, not registers.
int trunk_chain_capacity(int tile_rows) {
    static int cap[2] = {0, 0};
    int& c = cap[tile_rows == 64 ? 1 : 0];
    if (c == 0) {
        int nb = 0;
        const void* fn = tile_rows == 64 ? (const void*)ppo_trunk_kernel<XRL_ACT_LEAKY_RELU, 0, 64, 4, 2, true>
                                         : (const void*)ppo_trunk_kernel<XRL_ACT_LEAKY_RELU, 0, 32, 4, 2, true>;
        const size_t lds = tile_rows == 64 ? TrunkLds<64>::BYTES : TrunkLds<32>::BYTES;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, FUSED_THREADS, lds) != hipSuccess || nb < 1) nb = 1;
        if (nb > 1) nb = 1;                                 // (one workgroup per CU is all the launch counts on)
        c = nb * device_cu_count();
    }
    return c;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_ppo_trunk_chain_fits(int32_t M, int32_t tile_rows, int64_t P) {
    if (M <= 0 || (tile_rows != 32 && tile_rows != 64) || P <= 0 || (P & 3)) return 0;
    const int grid = 2 * ((M + tile_rows - 1) / tile_rows);
    const int n_vb = (int)((P / 4 + 63) / 64);
    return (grid >= n_vb && grid <= XRL_CHAIN_MAX_WGS && grid <= trunk_chain_capacity(tile_rows)) ? 1 : 0;
}

extern "C" int xrl_ppo_trunk_chained(const xrl_ppo_fused_t* pp, const xrl_opt_chain_t* oo, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr && oo != nullptr);
    const xrl_ppo_fused_t& p = *pp;
    const xrl_opt_chain_t& o = *oo;
    XRL_CHECK_ARG(p.params != nullptr && p.l0_fold_off > 0 && p.dist == 0 && p.D == 4 && p.A == 2);
    XRL_CHECK_ARG(p.f_obs && p.f_act && p.f_ret && p.f_adv && p.f_logp && p.idx && p.slabs && p.partials);
    XRL_CHECK_ARG(p.M > 0 && p.n_envs > 0 && p.T > 0);
    if (!g_fast_enabled_ppo || !ppo_trunk_eligible(p)) { set_error("xrl_ppo_trunk_chained: the network is not of the shared-trunk family (csrc/ppo_trunk.hip)"); return XRL_EINVAL; }
    XRL_CHECK_ARG(o.slabs && o.params && o.grad && o.m && o.v && o.state && o.sumsq_part && o.sync && o.n_split >= 1 && o.P > 0);
    XRL_CHECK_ARG(o.params == p.params);                      // the minibatch runs on the parameters the prologue leaves
    XRL_CHECK_ARG((o.P & 3) == 0 && (o.slab_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(o.slabs) & 15) == 0));
    const int n_vb = (int)((o.P / 4 + 63) / 64);
    XRL_CHECK_ARG(n_vb <= o.n_part && o.n_part <= 1024);
    const xrl_mirrors_t& mir = o.mirrors;
    XRL_CHECK_ARG(mir.n >= 0 && mir.n <= XRL_MAX_MIRRORS);
    for (int q = 0; q < mir.n; ++q) XRL_CHECK_ARG(mir.map[q] && mir.dst[q]);
    XRL_CHECK_ARG(mir.target == nullptr && mir.target_image == nullptr && mir.tick == nullptr && mir.part == nullptr && mir.part_out == nullptr && mir.alt_split == 0);
    XRL_CHECK_ARG(mir.split_plane == 0 && p.frag16 == nullptr);   // (the chained launch has no split-product instances and no split mirror stores)
    XRL_CHECK_ARG(mir.fold_len >= 0 && (mir.fold_len & 3) == 0 && (mir.fold_off & 3) == 0 &&
                  (mir.fold_len == 0 || (mir.fold_off >= o.P && mir.fold_off + mir.fold_len <= o.slab_stride && mir.fold_len <= o.P)));
    const int tile_rows = p.pad0 == 64 ? 64 : 32;
    if (!xrl_ppo_trunk_chain_fits(p.M, tile_rows, o.P)) {
        set_error("xrl_ppo_trunk_chained: %d rows in %d-row tiles for %lld parameters do not fit the chained launch on this device "
                  "(needs ceil(P / 256) <= workgroups <= resident capacity; xrl_ppo_trunk_chain_fits tells in advance)", p.M, tile_rows, (long long)o.P);
        return XRL_EINVAL;
    }
    return launch_ppo_trunk(p, &o, as_stream(stream));
}

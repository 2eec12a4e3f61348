// On-policy rollout of the two-branch Gaussian class D-256-256-{A | 1} (configs/ppo/mujoco.yaml: Basic_Identical, actor / critic
// hidden [256, 256], activation_action tanh) on the device-resident continuous-control provider (xrl_synth_control_step's
// dynamics), as ONE launch per rollout with only the ACTOR on the step chain -- the construction of csrc/rollout_actor.hip
// (CartPole class) at HalfCheetah shapes (BASELINE configs[3], 128 envs per GPU):
//   * workgroup w owns 16 envs; all eight waves are matrix waves (16x16x4 fp32 MFMA, everything computed transposed so that a
//     layer's result tile is the next product's B operand): wave v holds hidden units [32 v, 32 v + 32) of the 256 x 256 middle
//     layer in registers (128 VGPRs) for the whole rollout; the first layer's activations go through LDS once per step (every wave
//     needs all 256 of them), the mean's partial sums through LDS to the sampling wave;
//   * the vector step's dependency chain: observation statistics over all envs -> normalise -> actor -> Normal(mu, std).sample()
//     -> dynamics -> next observations.  Values, bootstrap values, reward normalisation and ret_rms.update consume a step without
//     feeding the next one: a trailing workgroup does the return statistics / rewards (as in rollout_actor.hip), the values of the
//     whole rollout are one batched forward pass afterwards (the caller's);
//   * per step the workgroups exchange their 16-row partial sums of the new raw observations (2 D doubles) as tagged 8-byte units
//     (data and flag in one message, two slots by step parity); wave v publishes, collects and merges dimensions v, v + 8, v + 16.
// The same kernel runs one launch per vector step (n_steps = 1: per-step callbacks, fallback), bit-identical.
// Reference semantics: ppo_agent.py:111-177, on_policy.py:128-169, actor_head.py:45-72, distributions.py:165-192,
// statistic_tools.py:117-185, agent.py:262-294.
#include "common.h"
#include "rng.h"

namespace xrl {

typedef float wf32x4 __attribute__((ext_vector_type(4)));

constexpr int WR = 16;                    // envs per workgroup
constexpr int WH = 256;                   // hidden width
constexpr int WLD = WH + 4;               // LDS row stride of the first layer's activations
constexpr int WDM = 20;                   // observation width limit (row stride of the small per-row arrays)
constexpr int WAM = 8;                    // action width limit
constexpr int WTH = 512;                  // threads per workgroup (8 waves, all matrix waves)
constexpr int WMAXWG = 16;
constexpr int WLC = 2;                    // k-chunks (of 16) of the middle layer's weights kept in LDS instead of registers
// exchange scratch (32-bit words): two slots of WSLOT 8-byte units -- unit ((d * 16 + wg) * 4 + k), k = lo s1, lo s2, hi s1, hi s2 of
// dimension d; progress words; XCC mask
// (measured, r04: spreading the dimensions over more L2 channels, 16-byte polling loads and a back-off between polling rounds
//  all leave the launch time unchanged -- the wait is the message's flight, not a queue)
constexpr int WUNITS = 16 * 4;
constexpr int WSLOT = WDM * WUNITS;
constexpr int WX_DONE = 4 * WSLOT, WX_MASK = WX_DONE + 32, WX_WORDS = WX_DONE + 64;

template <int CTRL>
__device__ __forceinline__ double wdpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_row16_sum(double v) {
    v += wdpp<0x128>(v); v += wdpp<0x124>(v); v += wdpp<0x122>(v); v += wdpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float w_ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned w_ld_dev(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long w_ld_dev(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void w_st_dev(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void w_st(T* p, T v, bool multi) { if (multi) w_st_dev(p, v); else *p = v; }

#define WMFMA(a, b, acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), acc, 0, 0, 0)

// the trailing workgroup: normalised rewards of step t (return statistics BEFORE that step's episode ends, ppo_agent.py:128), then
// ret_rms.update(returns[i:i+1]) for every env whose episode ended at step t, in env order (:146-149)
__device__ __forceinline__ void wide_bookkeeper(const xrl_rollout_wide_t& q, int n_act) {
#pragma clang fp contract(off)
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, n = q.n, n4 = (n + 3) & ~3;
    unsigned* xw = q.xchg;
    float mean = *q.ret_mean, var = *q.ret_var;
    double count = *q.ret_count;
    for (int k = 0; k < q.n_steps; ++k) {
        const int t = q.t0 + k;
        int spins = 0;
        bool dead = false;
        for (;;) {
            const unsigned f = lane < n_act ? w_ld_dev(xw + WX_DONE + lane) : 0xffffffffu;
            if (__ballot(f < (unsigned)(k + 1)) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255) == 0 && (spins > 2000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
        }
        if (dead) return;
        float rstd = sqrtf(var);
        rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
        for (int base = 0; base < n4; base += 256) {
            const int e4 = base + 4 * lane;
            unsigned w = 0u;
            float rf[4] = {0.f, 0.f, 0.f, 0.f};
            if (e4 < n4) {
                w = w_ld_dev(reinterpret_cast<const unsigned*>(q.ended + (size_t)t * n4 + e4));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    rf[b] = w_ld_dev(q.ret_final + (size_t)t * n4 + e4 + b);
                    if (e4 + b < n) {
                        const float r = w_ld_dev(q.raw_rew + (size_t)t * n4 + e4 + b);
                        float rn = r;
                        if (q.use_rewnorm) rn = fminf(fmaxf(r / rstd, -q.rew_range), q.rew_range);
                        q.f_rew[(size_t)t * n + e4 + b] = rn;
                    }
                }
            }
            unsigned long long mm = __ballot(w != 0u);
            while (mm) {
                const int src = __ffsll((long long)mm) - 1; mm &= mm - 1;
                const unsigned ws = (unsigned)__builtin_amdgcn_readlane((int)w, src);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float bmv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rf[b]), src));
                    if ((ws >> (8 * b)) & 0xffu) {
                        const double tot = count + 1.0; const float delta = bmv - mean;
                        const float new_mean = mean + delta * 1.0f / (float)tot;
                        const float M2 = var * (float)count + 0.f + (delta * delta) * (float)count * 1.0f / (float)tot;
                        mean = new_mean; var = M2 / (float)tot; count = tot;
                    }
                }
            }
        }
    }
    if (lane == 0) { *q.ret_mean = mean; *q.ret_var = var; *q.ret_count = count; }
}

// TAPE: the provider's outputs and the action draws' normals come from a recorded tape (xrl_rollout_wide_t.tape_*); every other
// instruction is shared with the timed instances.
template <int ACT, int OACT, bool TAPE>
__global__ void __launch_bounds__(WTH) wide_rollout_kernel(xrl_rollout_wide_t q) {
#pragma clang fp contract(off)
    if (blockIdx.x & 7) return;                                  // one XCD's share of the grid (rollout_actor.hip)
    const int n = q.n, n_act = (n + WR - 1) / WR;
    const int wg = blockIdx.x >> 3;
    if (wg >= n_act) { wide_bookkeeper(q, n_act); return; }

    __shared__ __attribute__((aligned(16))) float h1[WR * WLD];          // first layer's activations [row][unit]
    __shared__ __attribute__((aligned(16))) float s_raw[WR][WDM];        // raw observations the envs act on
    __shared__ __attribute__((aligned(16))) float s_st[WR][WDM];         // simulator state
    __shared__ __attribute__((aligned(16))) float s_norm[2][WDM];        // mean | 1 / (std + 1e-8) of the step
    __shared__ __attribute__((aligned(16))) float pmu[8][WR][WAM];       // partial pre-activations of the mean [wave][row][action]
    __shared__ __attribute__((aligned(16))) float s_act[WR][WAM];        // sampled actions of the step
    __shared__ __attribute__((aligned(16))) float s_zn[2][WR][WAM];      // standard normals of the action draw, by step parity
    __shared__ __attribute__((aligned(16))) float s_amatT[WDM][WDM], s_bmatT[WDM][WAM];   // provider matrices, transposed: [output][input]
    // the small parameters (first layer, both hidden biases, head rows): read from here every step -- only the 256 x 256 layer's
    // 128 fragment registers per lane stay resident (with the small ones in registers as well the kernel spilled 42 VGPRs)
    __shared__ float s_w0[WH * WDM], s_b0[WH], s_b1[WH], s_w2[WAM * WH];
    __shared__ __attribute__((aligned(16))) float s_w1l[8 * 2 * WLC * 64][4];      // [wave][tile][chunk][lane]: the middle layer's LDS-resident k-chunks
    __shared__ float s_head[3][WAM];                                     // head bias | std | log std per action
    __shared__ int s_fin[WR][2];                                         // episodes finished in this launch: count, steps
    __shared__ double s_fin_score[WR];
    __shared__ float s_y0[WR];
    __shared__ int s_trunc[WR];
    __shared__ int s_abort, s_multi;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, g = lane >> 4;
    const int D = q.D, A = q.A, n_steps = q.n_steps, T = q.T, t0 = q.t0;
    const int e0 = wg * WR;
    const bool use_norm = q.use_obsnorm != 0;
    const bool single = n_act == 1;
    const float obs_range = q.obs_range;
    const float* P = q.params;
    long long* dbg = q.dbg;
    const bool dbg_on = dbg != nullptr && wg == 0 && tid == 0;
    const int dbg_k = n_steps / 2;

    // ---- item threads: thread i < 16 D owns (row ir, dimension id) of the physics / records; per-row counters live with id == 0
    const int it_ = tid - 64;                                    // (items start at wave 1: wave 0 samples while they prepare / integrate)
    const bool item = it_ >= 0 && it_ < WR * D;
    const int ir = item ? it_ / D : 0, id = item ? it_ - ir * D : 0, ie = e0 + ir;
    const bool item_ok = item && ie < n;
    int ep_steps = 0;
    float ep_score = 0.f, rtrack = 0.f, row_pen = 0.f;
    // ---- one-time loads
    for (int i = tid; i < WDM * WDM; i += WTH) { const int o = i / WDM, kk = i - o * WDM; s_amatT[o][kk] = (o < D && kk < D) ? q.Amat[kk * D + o] : 0.f; }
    for (int i = tid; i < WDM * WAM; i += WTH) { const int o = i / WAM, j = i - o * WAM; s_bmatT[o][j] = (o < D && j < A) ? q.Bmat[j * D + o] : 0.f; }
    for (int i = tid; i < WR * WDM; i += WTH) {
        const int r = i / WDM, d = i - r * WDM;
        const bool ok = d < D && e0 + r < n;
        s_raw[r][d] = ok ? q.obs_raw[(size_t)(e0 + r) * D + d] : 0.f;
        s_st[r][d] = ok ? q.env_state[(size_t)(e0 + r) * D + d] : 0.f;
    }
    if (tid < 2 * WDM) s_norm[tid / WDM][tid % WDM] = tid < WDM ? 0.f : 1.f;
    // per-row episode counters: wave 0's lanes 0..15 (row = lane) -- the wave that samples also keeps the books of the rows
    const bool rowl = tid < WR, row_ok = rowl && e0 + tid < n;
    if (row_ok) { ep_steps = q.env_steps[e0 + tid]; ep_score = q.env_score[e0 + tid]; rtrack = q.ret_track[e0 + tid]; }
    if (tid == 0) { s_abort = 0; s_multi = (q.flags & 1) ? 1 : 0; }
    // weights: the middle layer's rows of this wave's hidden units [32 wave, +32) (two 16-unit tiles) in registers; the rest in LDS
    // (round 6: the last WLC of the 16 k-chunks live in LDS, not in registers -- 8 VGPRs per chunk: with all 128 fragment registers
    //  resident the kernel sat at the 256-register limit and spilled 7-10 VGPRs inside the step loop; two ds_read_b128 per chunk and
    //  step instead, in the shadow of the MFMAs)
    float4 w1f[2][16 - WLC];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u0 = 32 * wave + 16 * j;
#pragma unroll
        for (int c = 0; c < 16 - WLC; ++c) w1f[j][c] = *reinterpret_cast<const float4*>(P + q.w1 + (size_t)(u0 + cl) * WH + 16 * c + 4 * g);
#pragma unroll
        for (int c = 16 - WLC; c < 16; ++c)
            *reinterpret_cast<float4*>(&s_w1l[((wave * 2 + j) * WLC + (c - (16 - WLC))) * 64 + lane][0]) =
                *reinterpret_cast<const float4*>(P + q.w1 + (size_t)(u0 + cl) * WH + 16 * c + 4 * g);
    }
    for (int i = tid; i < WH * WDM; i += WTH) { const int u = i / WDM, d = i - u * WDM; s_w0[i] = d < D ? P[q.w0 + (size_t)u * D + d] : 0.f; }
    for (int i = tid; i < WH; i += WTH) { s_b0[i] = P[q.b0 + i]; s_b1[i] = P[q.b1 + i]; }
    for (int i = tid; i < WAM * WH; i += WTH) s_w2[i] = i < A * WH ? P[q.w2 + i] : 0.f;
    // statistics of this wave's dimensions d = wave, wave + 8, wave + 16 (DPP row g: dimension wave + 8 g; row 3 idles)
    const int sd_ = wave + 8 * g;
    const bool sd_ok = g < 3 && sd_ < D;
    float st_mean = 0.f, st_var = 1.f;
    double st_cnt = 0.0;
    if (use_norm && sd_ok) { st_mean = q.obs_mean[sd_]; st_var = q.obs_var[sd_]; st_cnt = *q.obs_count; }
    unsigned long long* xu = reinterpret_cast<unsigned long long*>(q.xchg);
    constexpr int ustride = WUNITS;
    const uint32_t pstep0 = q.step + (q.step_dev ? *q.step_dev : 0u) + (uint32_t)t0;        // Philox step of the policy's draws
    const uint32_t estep0 = q.env_step + (q.env_step_dev ? *q.env_step_dev : 0u) + (uint32_t)t0;   // ... of the simulator's noise
    if (tid < WAM) {
        const float sd = tid < A ? expf(P[q.log_std_off + tid]) : 1.f;
        s_head[0][tid] = tid < A ? P[q.b2 + tid] : 0.f; s_head[1][tid] = sd; s_head[2][tid] = logf(sd);
    }
    if (tid < WR) { s_fin[tid][0] = 0; s_fin[tid][1] = 0; s_fin_score[tid] = 0.0; }
    __syncthreads();

    // partial sums of this workgroup's rows for the wave's dimensions + the message carrying them (tag = step + 1)
    // (not kept across the step: several workgroups read their own contribution back from the exchange, one recomputes it in P1)
    auto partial_sums = [&](double& ps1, double& ps2) {
        const double v = (sd_ok && e0 + cl < n) ? (double)s_raw[cl][sd_] : 0.0;
        ps1 = w_row16_sum(v); ps2 = w_row16_sum(v * v);
    };
    auto publish = [&](unsigned tag, bool dev) {
        double ps1, ps2;
        partial_sums(ps1, ps2);
        if (sd_ok && cl < 4) {
            const double sv = (cl & 1) ? ps2 : ps1;
            const unsigned word = (cl & 2) ? (unsigned)__double2hiint(sv) : (unsigned)__double2loint(sv);
            const unsigned long long m = ((unsigned long long)word << 32) | (unsigned long long)tag;
            unsigned long long* dst = xu + (tag & 1u) * WSLOT + sd_ * ustride + wg * 4 + cl;
            if (dev) w_st_dev(dst, m); else *dst = m;
        }
    };
    // The normals of a step -- the simulator's noise (item threads) and the action draws (Philox stream of xrl_policy_sample; the
    // last two waves' threads i < 16 A, free of item threads for D <= 20) -- depend on nothing the step computes: one Philox call
    // per thread, written WITHOUT a branch so that the step loop can issue it between the middle layer's MFMAs (the matrix pipe
    // bounds that phase; the vector slots beside it are idle) one step ahead.
    const int di = tid - (WTH - WR * WAM);
    const bool drawer = di >= 0 && di < WR * A;
    const int dr = drawer ? di / A : 0, dj = drawer ? di - dr * A : 0;
    const uint64_t nz_seed = drawer ? q.seed : q.env_seed;
    const uint32_t nz_env = drawer ? (uint32_t)(e0 + dr) : (uint32_t)ie;
    const uint32_t nz_stream = drawer ? 0x47415500u + (uint32_t)dj : 0x53594E00u + (uint32_t)id;
    const uint32_t nz_step0 = drawer ? pstep0 : estep0;
    const float nz_floor = drawer ? 5.96e-8f : 1e-7f, nz_scale = drawer ? 1.f : 0.01f;     // (policy_normal | 0.01 provider_normal)
    const size_t tape_row0 = TAPE ? (size_t)(*q.tape_pos) + (size_t)t0 : 0;
    auto step_normal = [&](int k) -> float {
        if constexpr (TAPE) {                                    // supplied normals of the action draw; the provider's noise is on the tape
            if (q.tape_z) return (drawer && t0 + k < T && e0 + dr < n) ? q.tape_z[((size_t)(t0 + k) * n + e0 + dr) * A + dj] : 0.f;
        }
        uint32_t r[4];
        philox4x32(nz_seed, nz_env, nz_step0 + (uint32_t)k, nz_stream, r);
        const float u1 = fmaxf(u01(r[0]), nz_floor), u2 = u01(r[1]);
        return nz_scale * (__builtin_amdgcn_sqrtf(-2.f * __logf(u1)) * __builtin_amdgcn_cosf(u2));
    };
    // the state's part of the dynamics' pre-activation (sum over kk in order, as xrl_synth_control_step): behind publish(), in front
    // of the poll -- off the step chain
    float noise = 0.f, pre_s = 0.f;
    auto prepare = [&]() {
        if (item) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < WDM / 4; ++c) {
                const float4 sv = *reinterpret_cast<const float4*>(&s_st[ir][4 * c]);
                const float4 av = *reinterpret_cast<const float4*>(&s_amatT[id][4 * c]);
                if (4 * c + 0 < D) acc += sv.x * av.x;
                if (4 * c + 1 < D) acc += sv.y * av.y;
                if (4 * c + 2 < D) acc += sv.z * av.z;
                if (4 * c + 3 < D) acc += sv.w * av.w;
            }
            pre_s = acc;
        }
    };
    bool multi = true;                                           // until the placement is known: device-scope message stores
    if (use_norm) {
        if (!single) {
            if (tid == 0) {
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                if (wg == 0) __hip_atomic_store(q.status + 1, (int)(xcc & 0xf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(q.status + 2, 1 << (xcc & 0xf));
                const unsigned seen = atomicOr(q.xchg + WX_MASK, 1u << (xcc & 0xf));
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
            }
            __syncthreads();                                     // (every wave's first message behind the workgroup's XCC bit)
            publish(1u, true);
        }
    }
    {
        const float z0 = step_normal(0);
        if (drawer) s_zn[0][dr][dj] = z0;
        noise = z0;
    }
    prepare();

    int k = 0;
    for (; k < n_steps; ++k) {
        const int t = t0 + k;
        const bool stamp = dbg_on && k == dbg_k;
        if (stamp) dbg[0] = clock64();
        const bool wstamp = dbg != nullptr && wg == 0 && lane == 0 && k == dbg_k;
        if (wstamp) dbg[16 + wave] = clock64();                  // (this wave's prepare() is done)
        // ================= P1: statistics of the step (every wave: its dimensions)
        if (use_norm) {
            double S1 = 0.0, S2 = 0.0;
            if (single) partial_sums(S1, S2);
            const double cnt = st_cnt, tot = cnt + (double)n;
            const float nf = (float)n, cntf = (float)cnt, rt = 1.0f / (float)tot;
            if (!single) {
                const unsigned tag = (unsigned)(k + 1);
                const bool live = sd_ok && cl < n_act;
                const unsigned long long* xs = xu + (tag & 1u) * WSLOT + (sd_ok ? sd_ : 0) * ustride + cl * 4;
                unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                int spins = 0;
                for (;;) {
                    if (live) { u0 = w_ld_dev(xs); u1 = w_ld_dev(xs + 1); u2 = w_ld_dev(xs + 2); u3 = w_ld_dev(xs + 3); }
                    const bool ok = !live || ((unsigned)u0 == tag && (unsigned)u1 == tag && (unsigned)u2 == tag && (unsigned)u3 == tag);
                    if (__ballot(!ok) == 0ull) break;
                    if ((++spins & 255) == 0 && (spins > 2000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        if (lane == 0) { __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
                        break;
                    }
                }
                if (wstamp) { dbg[24 + wave] = clock64(); dbg[32 + wave] = spins; }      // (its messages are in)
                const double v1 = live ? __hiloint2double((int)(u2 >> 32), (int)(u0 >> 32)) : 0.0;
                const double v2 = live ? __hiloint2double((int)(u3 >> 32), (int)(u1 >> 32)) : 0.0;
                S1 = w_row16_sum(v1); S2 = w_row16_sum(v2);
                if (k == 0) {
                    const unsigned mask = w_ld_dev(q.xchg + WX_MASK);
                    multi = __popc(mask) != 1 || (q.flags & 1);
                    if (tid == 0 && multi && wg == 0) atomicAdd(q.status + 3, 1);
                }
            }
            // RunningMeanStd.update for dimension sd_ (the arithmetic of rollout_actor.hip's merge)
            if (sd_ok) {
                const bool pow2 = (n & (n - 1)) == 0;
                const double inv_n = 1.0 / (double)n;
                const double m = pow2 ? S1 * inv_n : S1 / n;
                const float bmean = (float)m;
                const float bv = (float)fmax((pow2 ? S2 * inv_n : S2 / n) - m * m, 0.0);
                const float delta = bmean - st_mean;
                const float new_mean = st_mean + delta * nf * rt;
                const float M2 = st_var * cntf + bv * nf + (delta * delta) * cntf * nf * rt;
                const float new_var = M2 * rt;
                st_mean = new_mean; st_var = new_var; st_cnt = tot;
                if (cl == 0) { s_norm[0][sd_] = new_mean; s_norm[1][sd_] = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(new_var) + 1e-8f); }
            }
        }
        if (stamp) dbg[1] = clock64();
        if (wstamp) dbg[40 + wave] = clock64();
        lds_barrier();                                                                             // #1 statistics ready
        if (s_abort) break;
        // ================= P2: first layer (wave: its 32 units), transposed: D[unit][row] = sum_d W0[unit][d] xn[row][d]
        {
            float xn[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int d = min(4 * c + g, WDM - 1);
                float x = s_raw[cl][d];
                if (use_norm) x = fminf(fmaxf((x - s_norm[0][d]) * s_norm[1][d], -obs_range), obs_range);
                xn[c] = x;
            }
            if (item_ok) {                                       // memory.observations[t]
                float x = s_raw[ir][id];
                if (use_norm) x = fminf(fmaxf((x - s_norm[0][id]) * s_norm[1][id], -obs_range), obs_range);
                q.f_obs[((size_t)t * n + ie) * D + id] = x;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u0 = 32 * wave + 16 * j;
                const float4 bb = *reinterpret_cast<const float4*>(&s_b0[u0 + 4 * g]);
                wf32x4 a = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int c = 0; c < 5; ++c) WMFMA(s_w0[(u0 + cl) * WDM + 4 * c + g], xn[c], a);       // A[m = unit][k = obs dim]
                float4 o;
                o.x = act_apply_c<ACT>(a[0]); o.y = act_apply_c<ACT>(a[1]); o.z = act_apply_c<ACT>(a[2]); o.w = act_apply_c<ACT>(a[3]);
                *reinterpret_cast<float4*>(h1 + cl * WLD + 32 * wave + 16 * j + 4 * g) = o;
            }
        }
        lds_barrier();                                                                             // #2 h1 ready
        if (stamp) dbg[2] = clock64();
        // ================= P3: middle layer (this wave's 32 units over all 256 inputs) + its share of the mean's pre-activation
        float nz_next;
        {
            wf32x4 acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 bb = *reinterpret_cast<const float4*>(&s_b1[32 * wave + 16 * j + 4 * g]);
                acc[j][0] = (wf32x4){bb.x, bb.y, bb.z, bb.w}; acc[j][1] = (wf32x4){0.f, 0.f, 0.f, 0.f};
            }
            const float* hrow = h1 + cl * WLD + 4 * g;
            nz_next = step_normal(k + 1);                        // (vector work for the slots between the MFMAs below)
#pragma unroll
            for (int c4 = 0; c4 < 16; c4 += 4) {
                float4 hf[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) hf[c] = *reinterpret_cast<const float4*>(hrow + 16 * (c4 + c));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cc = c4 + c, par = c & 1;
                    float4 wa_, wb_;
                    if (cc < 16 - WLC) { wa_ = w1f[0][cc < 16 - WLC ? cc : 0]; wb_ = w1f[1][cc < 16 - WLC ? cc : 0]; }
                    else {
                        wa_ = *reinterpret_cast<const float4*>(&s_w1l[((wave * 2 + 0) * WLC + (cc - (16 - WLC))) * 64 + lane][0]);
                        wb_ = *reinterpret_cast<const float4*>(&s_w1l[((wave * 2 + 1) * WLC + (cc - (16 - WLC))) * 64 + lane][0]);
                    }
                    WMFMA(wa_.x, hf[c].x, acc[0][par]); WMFMA(wb_.x, hf[c].x, acc[1][par]);
                    WMFMA(wa_.y, hf[c].y, acc[0][par]); WMFMA(wb_.y, hf[c].y, acc[1][par]);
                    WMFMA(wa_.z, hf[c].z, acc[0][par]); WMFMA(wb_.z, hf[c].z, acc[1][par]);
                    WMFMA(wa_.w, hf[c].w, acc[0][par]); WMFMA(wb_.w, hf[c].w, acc[1][par]);
                }
            }
            // (pin the pattern: one MFMA, then up to three of the normal's vector instructions, while those last)
#pragma unroll
            for (int i = 0; i < 44; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
            __builtin_amdgcn_sched_barrier(0);
            wf32x4 lg = {0.f, 0.f, 0.f, 0.f}, lg2 = {0.f, 0.f, 0.f, 0.f};
            // head: A[m = action cl][k = unit 4 g + i] from the LDS copy (rows >= A are zero)
            const float4 wa = *reinterpret_cast<const float4*>(&s_w2[min(cl, WAM - 1) * WH + 32 * wave + 4 * g]);
            const float4 wb = *reinterpret_cast<const float4*>(&s_w2[min(cl, WAM - 1) * WH + 32 * wave + 16 + 4 * g]);
            const float wz = cl < WAM ? 1.f : 0.f;
            const float wha[4] = {wa.x * wz, wa.y * wz, wa.z * wz, wa.w * wz}, whb[4] = {wb.x * wz, wb.y * wz, wb.z * wz, wb.w * wz};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                WMFMA(wha[i], act_apply_c<ACT>(acc[0][0][i] + acc[0][1][i]), lg);
                WMFMA(whb[i], act_apply_c<ACT>(acc[1][0][i] + acc[1][1][i]), lg2);
            }
            // lg[i] of lane (g, cl) = partial pre-activation of action 4 g + i, row cl
            asm volatile("" : "+v"(nz_next));                    // (keeps the normal's instructions in THIS block: the compiler sinks them to the use)
            if (g < 2) *reinterpret_cast<float4*>(&pmu[wave][cl][4 * g]) = make_float4(lg[0] + lg2[0], lg[1] + lg2[1], lg[2] + lg2[2], lg[3] + lg2[3]);
        }
        if (drawer) s_zn[(k + 1) & 1][dr][dj] = nz_next;           // (read in P4 of step k + 1; P4 of this step reads the other parity)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores of the previous step are in L2 (long since)
        lds_barrier();                                                                             // #3 partial means ready
        if (tid == 64 && k > 0) w_st_dev(q.xchg + WX_DONE + wg, (unsigned)k);       // steps < k complete (trailing readers)
        if (stamp) dbg[3] = clock64();
        // ================= P4: wave 0: Normal(mu, std).sample(), log-prob
        if (wave == 0) {
            // lane (g, row): actions j = g and g + 4
            float lp = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = g + 4 * h;
                if (j < A) {
                    float pre = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) pre += pmu[w][cl][j];
                    const float mu = act_apply_c<OACT>(pre + s_head[0][j]);
                    const float sd = s_head[1][j];
                    const float x = mu + sd * s_zn[k & 1][cl][j];                                // Normal(mu, std).sample()
                    const float df = x - mu;
                    lp += -(df * df) / (2.f * sd * sd) - s_head[2][j] - 0.91893853320467274178f;
                    s_act[cl][j] = x;
                    if (e0 + cl < n) q.f_act[((size_t)t * n + e0 + cl) * A + j] = x;
                }
            }
            lp += __shfl_xor(lp, 16, 64); lp += __shfl_xor(lp, 32, 64);
            if (g == 0 && e0 + cl < n) q.f_logp[(size_t)t * n + e0 + cl] = lp;
            if (rowl) {                                          // the row's episode-end flags (bit 0 over, bit 1 terminated) and action penalty (sum over j in order)
                if constexpr (TAPE) {
                    const size_t r = tape_row0 + (size_t)k;
                    const bool okr = e0 + tid < n && r < (size_t)q.tape_rows;
                    const bool tm = okr && q.tape_term[r * n + e0 + tid] > 0.f, tc = okr && q.tape_trunc[r * n + e0 + tid] > 0.f;
                    s_trunc[tid] = ((tm || tc) ? 1 : 0) | (tm ? 2 : 0);
                } else
                s_trunc[tid] = (ep_steps + 1 >= q.max_steps) ? 1 : 0;
                float pen = 0.f;
#pragma unroll
                for (int j = 0; j < WAM; ++j) if (j < A) { const float ai = fminf(fmaxf(s_act[tid][j], -1.f), 1.f); pen += ai * ai; }
                row_pen = pen;
            }
        }
        lds_barrier();                                                                             // #4 actions ready
        if (stamp) dbg[4] = clock64();
        // ================= P5: dynamics (item threads), next raw observations / state
        if (item) {
            float acc = pre_s;
            const float4 a0 = *reinterpret_cast<const float4*>(&s_act[ir][0]), a1 = *reinterpret_cast<const float4*>(&s_act[ir][4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&s_bmatT[id][0]), b1 = *reinterpret_cast<const float4*>(&s_bmatT[id][4]);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < WAM; ++j) if (j < A) acc += fminf(fmaxf(av[j], -1.f), 1.f) * bv[j];
            float y = provider_tanh(acc) + noise;
            if constexpr (TAPE) {                                // what the recorded simulator returned
                const size_t r = tape_row0 + (size_t)k;
                y = (item_ok && r < (size_t)q.tape_rows) ? q.tape_next_obs[(r * n + ie) * D + id] : 0.f;
            }
            if (id == 0) s_y0[ir] = y;
            // next observation before the reset, normalised with this step's statistics (get_terminated_values' input)
            if (item_ok) {
                float xv = y;
                if (use_norm) xv = fminf(fmaxf((xv - s_norm[0][id]) * s_norm[1][id], -obs_range), obs_range);
                q.xnext[((size_t)t * n + ie) * D + id] = xv;
            }
            float v = y;
            if constexpr (TAPE) {                                // infos[i]["reset_obs"] of the recorded step
                const size_t r = tape_row0 + (size_t)k;
                if (s_trunc[ir] != 0) v = (item_ok && r < (size_t)q.tape_rows) ? q.tape_reset_obs[(r * n + ie) * D + id] : 0.f;
            } else
            if (s_trunc[ir] != 0) v = 0.1f * provider_normal(q.env_seed, (uint32_t)ie, estep0 + (uint32_t)k, 0x53594E00u + 64u + (uint32_t)id);
            s_st[ir][id] = v; s_raw[ir][id] = v;
        }
        noise = nz_next;
        lds_barrier();                                                                             // #5 next raw rows ready
        if (stamp) dbg[5] = clock64();
        if (use_norm && !single && k + 1 < n_steps) publish((unsigned)(k + 2), multi);
        if (k + 1 < n_steps) prepare();
        // the rows' records and episode counters (wave 0's lanes 0..15; its stores trail the message)
        if (rowl) {
            const bool trunc = s_trunc[tid] != 0;                 // (the episode is over: truncated, or -- tape -- terminated)
            const bool term = TAPE && (s_trunc[tid] & 2) != 0;
            float rew = s_y0[tid] - 0.1f * row_pen;
            if constexpr (TAPE) {
                const size_t r = tape_row0 + (size_t)k;
                rew = (row_ok && r < (size_t)q.tape_rows) ? q.tape_rew[r * n + e0 + tid] : 0.f;
            }
            const int steps = ep_steps + 1;
            const float score = ep_score + rew;
            const float tr = q.gamma * rtrack + rew;              // self.returns = gamma * self.returns + rewards
            if (row_ok) {
                const int n4 = (n + 3) & ~3;
                const size_t o = (size_t)t * n + e0 + tid, o4 = (size_t)t * n4 + e0 + tid;
                q.f_term[o] = term ? 1.f : 0.f;
                q.f_seg[o] = (trunc || t == T - 1) ? (uint8_t)(1 | (term ? 6 : 0)) : (uint8_t)0;
                w_st_dev(q.raw_rew + o4, rew);
                if (trunc) w_st_dev(q.ret_final + o4, tr);
                w_st_dev(q.ended + o4, (uint8_t)(trunc ? 1 : 0));
                if (trunc) { s_fin[tid][0] += 1; s_fin[tid][1] += steps; s_fin_score[tid] += (double)score; }
            }
            ep_steps = trunc ? 0 : steps; ep_score = trunc ? 0.f : score; rtrack = trunc ? 0.f : tr;
        }
        if (stamp) { dbg[6] = clock64(); dbg[15] = 7; }
    }

    // ================= hand the state back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!s_abort && k == n_steps) {
        // (the hand-back pointers are read from the argument segment HERE: as by-value arguments they would occupy scalar registers
        //  through the whole step loop, and the loop already spills)
        typedef const __attribute__((address_space(4))) xrl_rollout_wide_t* QArgs;
        QArgs qa = (QArgs)__builtin_amdgcn_kernarg_segment_ptr();
        if (item_ok) {
            qa->obs_raw[(size_t)ie * D + id] = s_raw[ir][id];
            qa->env_state[(size_t)ie * D + id] = s_st[ir][id];
        }
        if (row_ok) {
            const int e = e0 + tid;
            qa->env_steps[e] = ep_steps; qa->env_score[e] = ep_score; qa->ret_track[e] = rtrack;
            if (s_fin[tid][0]) { atomicAdd(&qa->env_stats[0], (double)s_fin[tid][0]); atomicAdd(&qa->env_stats[1], s_fin_score[tid]); atomicAdd(&qa->env_stats[2], (double)s_fin[tid][1]); }
        }
        if (wg == 0 && use_norm && sd_ok && cl == 0) {
            qa->obs_mean[sd_] = st_mean; qa->obs_var[sd_] = st_var;
            if (sd_ == 0) *qa->obs_count = st_cnt;
        }
        if (tid == 0) w_st_dev(qa->xchg + WX_DONE + wg, (unsigned)n_steps);
    }
}

__global__ void copy_column_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i * ld];
}

__global__ void zero_wide_xchg_kernel(uint32_t* p) {
    for (int i = blockIdx.x * 512 + threadIdx.x; i < WX_WORDS; i += 512 * gridDim.x) p[i] = 0u;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_rollout_wide_words(void) { return WX_WORDS; }

extern "C" int xrl_copy_column(const float* src, int ld, float* dst, int64_t n, xrl_stream_t stream) {
    XRL_CHECK_ARG(src && dst && ld >= 1 && n >= 0);
    if (n == 0) return XRL_OK;
    hipLaunchKernelGGL(copy_column_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), src, ld, dst, n);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_rollout_wide_max_envs(void) {
    const int wg = device_cu_count() / 8 - 1;                           // (as xrl_rollout_cartpole_max_envs: WR envs per workgroup)
    return WR * (wg > 0 ? wg : 0);
}

extern "C" int xrl_rollout_wide_run(const xrl_rollout_wide_t* qq, xrl_stream_t stream) {
    XRL_CHECK_ARG(qq != nullptr);
    const xrl_rollout_wide_t& q = *qq;
    XRL_CHECK_ARG(q.params && q.n > 0 && q.n <= WR * WMAXWG && q.T >= 1 && q.t0 >= 0 && q.n_steps >= 1 && q.t0 + q.n_steps <= q.T);
    XRL_CHECK_ARG(q.D >= 1 && q.D <= WDM && q.A >= 1 && q.A <= WAM && q.H == WH);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(q.params) & 15) == 0 && (q.w1 & 3) == 0);
    XRL_CHECK_ARG(q.obs_raw && q.obs_mean && q.obs_var && q.obs_count && q.ret_mean && q.ret_var && q.ret_count && q.ret_track);
    XRL_CHECK_ARG(q.env_state && q.env_steps && q.env_score && q.env_stats && q.Amat && q.Bmat);
    XRL_CHECK_ARG(q.f_obs && q.f_act && q.f_logp && q.f_rew && q.f_term && q.f_seg && q.xnext && q.ended && q.ret_final && q.raw_rew);
    XRL_CHECK_ARG(q.xchg && q.status && (reinterpret_cast<uintptr_t>(q.ended) & 3) == 0 && (reinterpret_cast<uintptr_t>(q.xchg) & 7) == 0);
    XRL_CHECK_ARG(q.out_act == XRL_ACT_NONE || q.out_act == XRL_ACT_TANH);
    const int n_wg = (q.n + WR - 1) / WR + 1;                           // actors + the bookkeeper
    XRL_CHECK_ARG(n_wg <= device_cu_count() / 8);
    hipLaunchKernelGGL(zero_wide_xchg_kernel, dim3(16), dim3(512), 0, as_stream(stream), q.xchg);
    if (q.tape_next_obs) {
        XRL_CHECK_ARG(q.tape_reset_obs && q.tape_rew && q.tape_term && q.tape_trunc && q.tape_pos && q.tape_rows >= 1);
        XRL_ACT_DISPATCH(q.act,
            if (q.out_act == XRL_ACT_TANH) hipLaunchKernelGGL((wide_rollout_kernel<ACT, XRL_ACT_TANH, true>), dim3(8 * n_wg), dim3(WTH), 0, as_stream(stream), q);
            else hipLaunchKernelGGL((wide_rollout_kernel<ACT, XRL_ACT_NONE, true>), dim3(8 * n_wg), dim3(WTH), 0, as_stream(stream), q);)
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    XRL_CHECK_ARG(q.tape_z == nullptr);                                  // (supplied normals ride with a tape only)
    XRL_ACT_DISPATCH(q.act,
        if (q.out_act == XRL_ACT_TANH) hipLaunchKernelGGL((wide_rollout_kernel<ACT, XRL_ACT_TANH, false>), dim3(8 * n_wg), dim3(WTH), 0, as_stream(stream), q);
        else hipLaunchKernelGGL((wide_rollout_kernel<ACT, XRL_ACT_NONE, false>), dim3(8 * n_wg), dim3(WTH), 0, as_stream(stream), q);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

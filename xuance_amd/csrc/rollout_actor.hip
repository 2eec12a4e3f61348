// On-policy rollout of the 4-128-{128-2,128-1} class on a device-resident CartPole, second design (round 4).
//
// What a vector step of PPO_Agent.train (ppo_agent.py:111-177, on_policy.py:128-169) REALLY chains: obs statistics over
// all envs -> normalise -> ACTOR forward -> sample -> envs.step -> next observations.  The critic (values of the stored
// observations, bootstrap values of truncated paths) and the return statistics (reward normalisation) consume a step's
// results but nothing of the next step depends on them.  So
//   * actor_rollout_kernel keeps ONLY the actor on the step chain: workgroup w owns 16 envs (one 16-row MFMA tile,
//     v_mfma_f32_16x16x4_f32: 64 instructions of 8 passes per matrix wave instead of the 64 x 16 passes of a 32-row tile),
//     weights in registers for all the steps of the launch, simulator state in LDS / registers;
//   * the only thing the workgroups exchange per step are their 16-row partial sums of the new raw observations (8 doubles).
//     They travel as sixteen 8-byte units {tag, half a double}: a unit is one atomic 64-bit store / load, the tag is the
//     step, so DATA AND FLAG ARE ONE MESSAGE -- a step boundary is one store latency + one load latency (the first design:
//     stores -> vmcnt(0) -> workgroup barrier -> flag store -> flag poll -> workgroup barrier -> data loads);
//   * one extra workgroup ("bookkeeper", one wave) trails the actors through per-workgroup progress words and does what
//     couples the envs without feeding back: ret_rms.update() of finished episodes in env order, the normalised reward;
//   * critic_values_kernel evaluates V(obs[t][e]) for the whole segment and V(next_obs) where a path was cut without
//     termination, AFTER the steps, as a batched pass over all 256 CUs (32-row tiles, first layer of tile i + 1 on the
//     vector waves while the matrix waves run tile i).
// The same actor kernel serves one launch per rollout (n_steps = T) and one launch per vector step (n_steps = 1, the mode
// for per-step callbacks and the fallback when the resident form is not available): state is handed over in memory, the
// arithmetic is the same instruction stream, results are bit-identical (tested).
// n <= 16: one workgroup, no exchange at all.  n <= 256: up to 16 workgroups on ONE XCD (grid 8x oversubscribed, only
// blockIdx % 8 == 0 stays, as in the first design: plain stores + device-scope loads are coherent inside one L2; the XCC
// ids are checked in every launch and the exchange falls back to device-scope stores when they differ).
#include "common.h"
#include "rng.h"
#include "cartpole.h"
#include "split3.h"

namespace xrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AR = 16;                    // envs (rows) per actor workgroup
constexpr int AH = 128;                   // hidden width
constexpr int ALD = AH + 4;               // LDS row stride of an activation tile (values kernel)
constexpr int ATH = 512;                  // threads per workgroup
constexpr int AMAXWG = 16;                // actor workgroups (16 x 16 = 256 envs)
// exchange scratch (32-bit words): [0, 1024) = two slots of 256 units of 8 bytes, the message of step k in slot k & 1, unit u of
// workgroup w at 64-bit index slot * 256 + u * 16 + w (a workgroup can only overwrite a slot two steps later, and it gets there
// only after every workgroup has published the step in between, i.e. has consumed the slot's old content);
// [1024, 1040) progress words; [1056] XCC mask of the launch
constexpr int XW_DONE = 1024, XW_MASK = 1056, XW_WORDS = 2048;

template <int CTRL>
__device__ __forceinline__ float adpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double adpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (rotations: every lane ends with the same bits)
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
    v += adpp<0x128>(v); v += adpp<0x124>(v); v += adpp<0x122>(v); v += adpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float a_ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned a_ld_dev(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long a_ld_dev(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void a_st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void a_st_dev(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void a_st_dev(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void a_st_dev(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a store another workgroup of the launch reads: plain inside one L2, device scope otherwise
template <typename T>
__device__ __forceinline__ void a_st(T* p, T v, bool multi) { if (multi) a_st_dev(p, v); else *p = v; }

#define MFMA16(a, b, acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), acc, 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------------
// the trailing workgroup: rewards of step t (normalised with the return statistics BEFORE that step's episode ends,
// ppo_agent.py:128), then ret_rms.update(returns[i:i+1]) for every env that finished at step t, in env order (:146-149)
__device__ __forceinline__ void rollout_bookkeeper(const xrl_rollout_run_t& q, int n_act) {
#pragma clang fp contract(off)
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, n = q.n, n4 = (n + 3) & ~3;
    unsigned* xw = q.xchg;
    float mean = q.ret_stats[0], var = q.ret_stats[1];
    double count = *q.ret_count;
    bool dead = false;
    for (int k = 0; k < q.n_steps && !dead; ++k) {
        const int t = q.t0 + k;
        float rstd = sqrtf(var);
        rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
        float rn = 1.0f;
        if (q.use_rewnorm) rn = fminf(fmaxf(1.0f / rstd, -q.rew_range), q.rew_range);
        for (int e = lane; e < n; e += 64) q.f_rew[(size_t)t * n + e] = rn;
        // every actor workgroup has completed step k (its stores of that step are in L2)
        int spins = 0;
        for (;;) {
            const unsigned f = lane < n_act ? a_ld_dev(xw + XW_DONE + lane) : 0xffffffffu;
            if (__ballot(f < (unsigned)(k + 1)) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255) == 0 && (spins > 2000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
        }
        if (dead) break;
        unsigned w = 0u;
        float rf[4] = {0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < n4; base += 256) {                 // (n <= 256: one round)
            const int e4 = base + 4 * lane;
            w = 0u;
            if (e4 < n4) {
                w = a_ld_dev(reinterpret_cast<const unsigned*>(q.ended + (size_t)t * n4 + e4));
#pragma unroll
                for (int b = 0; b < 4; ++b) rf[b] = a_ld_dev(q.ret_final + (size_t)t * n4 + e4 + b);
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
    if (lane == 0 && !dead) { q.ret_stats[0] = mean; q.ret_stats[1] = var; *q.ret_count = count; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Every wave role runs its OWN step loop (same three LDS barriers per step in each): what a role keeps in scalar registers
// is then live only on that role's path -- one loop body with all roles inside kept every kernel argument live through the
// chain wave's tail and restored ~110 of them per step from spill lanes.
// TAPE: the simulators' outputs (and, optionally, the sampling uniforms) come from a recorded tape (xrl_rollout_run_t.tape_*):
// the physics wave, the reset wave and the uniform draw READ what they otherwise compute; every other instruction is shared.
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
#define AMFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0)

// BX: the branch layer's 128-deep products as exact 3-way bf16 splits (csrc/ppo_trunk_bx.hip, csrc/split3.h) on v_mfma_f32_16x16x32_bf16:
// a k-step of 32 = the chunk pair (2 t, 2 t + 1); lane (g, cl) supplies, for A and for B alike, the eight k-values 32 t + 4 g + s and
// 32 t + 16 + 4 g + s (s = 0..3) -- a sum over k allows any assignment both operands share, so the float32 registers of the 16x16x4
// form (the weights' float4s, the first layer's result fragments) become the bf16 operands as they are.  The weights are split once per
// launch (they stay in registers for all its steps), the 32 activations of a lane once per step: 48 instructions of 16 pipe cycles per
// wave and step instead of 128 of 32.
template <int ACT, bool TAPE, bool BX = false>
__global__ void __launch_bounds__(ATH) actor_rollout_kernel(xrl_rollout_run_t q) {
#pragma clang fp contract(off)
    if (blockIdx.x & 7) return;                                  // keep one XCD's share of the grid (see header)
    const int n = q.n, n_act = (n + AR - 1) / AR;
    const int wg = blockIdx.x >> 3;
    if (wg >= n_act) { rollout_bookkeeper(q, n_act); return; }

    __shared__ __attribute__((aligned(16))) float s_raw[AR][4];          // raw observations this workgroup's envs act on
    __shared__ __attribute__((aligned(16))) float s_norm[8];             // mean[4] | 1 / (std[4] + 1e-8) after the step's statistics update
    __shared__ __attribute__((aligned(16))) float plog[4][AR][2];        // partial logits [matrix wave][row][action]
    __shared__ __attribute__((aligned(16))) double ph_state[2][AR][4];   // physics step for both actions (private to wave 5)
    __shared__ __attribute__((aligned(16))) float ph_f[2][2][AR][4];     // ... as the float32 observations they produce, by step parity
    __shared__ int ph_term[2][2][AR];
    __shared__ __attribute__((aligned(16))) double rs_state[2][AR][4];   // state after an auto-reset, by step parity
    __shared__ __attribute__((aligned(16))) float rs_f[2][AR][4];        // ... as float32 observations
    __shared__ __attribute__((aligned(16))) float s_rec[AR][4];          // the step's record of a row: action, log-prob, flags (bit 0 terminated, bit 1 episode over), return tracker
    __shared__ int s_sel[AR];                                            // what the step did to the env: 0 / 1 action taken, 2 reset
    __shared__ int ep_lds[AR];
    __shared__ float s_u[AR];
    __shared__ int s_abort, s_multi;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e0 = wg * AR, n_steps = q.n_steps;
    const bool use_norm = q.use_obsnorm != 0;
    const float obs_range = q.obs_range;
    const float* P = q.params;
    const int cl = lane & 15, g = lane >> 4;                     // lane = (DPP row g, position cl)
    long long* dbg = q.dbg;
    const bool dbg_on = dbg != nullptr && wg == 0 && lane == 0;
    const int dbg_k = n_steps / 2;

    // _process_observation (agent.py:262-280) of a raw row with the step's statistics (the quotient as a product with the
    // reciprocal the statistics wave took once: within 2 ulp of the division)
    auto normalise = [&](float4 x) -> float4 {
        if (use_norm) {
            const float4 nm = *reinterpret_cast<const float4*>(&s_norm[0]), ni = *reinterpret_cast<const float4*>(&s_norm[4]);
            x.x = fminf(fmaxf((x.x - nm.x) * ni.x, -obs_range), obs_range);
            x.y = fminf(fmaxf((x.y - nm.y) * ni.y, -obs_range), obs_range);
            x.z = fminf(fmaxf((x.z - nm.z) * ni.z, -obs_range), obs_range);
            x.w = fminf(fmaxf((x.w - nm.w) * ni.w, -obs_range), obs_range);
        }
        return x;
    };
    // A step of a workgroup, seen from its LDS barriers (all eight waves pass the same three per step):
    //   #1  statistics of the step (chain wave) and raw rows (records wave) in LDS  -> matrix waves: the whole actor network
    //   #3  partial logits in LDS                                                    -> chain wave: sample, the env's fate
    //   #0  fate + record scalars of the step in LDS  -> wave 5: physics of the NEXT step for both actions; wave 6: its reset draw;
    //       wave 7: this step's records to memory, next raw rows, next sampling uniform -- while the chain wave reduces, publishes
    //       and collects the messages of the next step.
    // (Helper work used to sit beside the matrix waves: a wave sharing a SIMD with back-to-back MFMAs gets ~1 VALU issue per MFMA,
    //  and with the 4 k-cycle chain of the 32-row tiles halved the helpers had become the longest waves of the step.)

    if (wave < 4) {
        // ================================================================ matrix waves: the actor network, registers to registers
        // Everything is computed TRANSPOSED, D[unit][row], because the 16x16x4 result layout -- lane (g, cl) holds D[4 g + i][cl] --
        // IS the B-operand layout of the next product over those 16 units as its k: lane (g, cl) supplies B[k = 4 g + s][n = cl] to
        // the MFMA with component s.  So
        //   first layer : 8 MFMAs (one per 16 hidden units = one k-chunk of the branch layer), A = W0 rows, B = the normalised
        //                 observation element x[row cl][dim g], C = bias; activation in place -> hf[c] (every matrix wave computes
        //                 all of it: no h1 in LDS, no barrier between the layers, the activations issue under the MFMAs in flight)
        //   branch layer: wave w owns hidden units [32 w, 32 w + 32) as two tiles; A = W1 straight from the row-major weights (lane
        //                 (g, cl): W1[col0 + cl][16 c + 4 g .. + 3]), B = hf[c]; four independent accumulator chains (even / odd
        //                 k-chunks per tile), the bias rides in as C
        //   logits      : 8 MFMAs with A = head weights (rows 0, 1 of a 16-row operand, the rest zero), B = the activated tile
        float4 wfr[2][8];
        float w0f[8];
        f32x4 b0f[8], bmf[2];
        float whf[2][4];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            w0f[c] = P[q.w0 + (size_t)(16 * c + cl) * 4 + g];                              // A[m = cl (hidden unit)][k = g (obs dim)]
#pragma unroll
            for (int i = 0; i < 4; ++i) b0f[c][i] = P[q.b0 + 16 * c + 4 * g + i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col0 = 32 * wave + 16 * j;
#pragma unroll
            for (int c = 0; c < 8; ++c) wfr[j][c] = *reinterpret_cast<const float4*>(P + q.w1 + (size_t)(col0 + cl) * AH + 16 * c + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bmf[j][i] = P[q.b1 + col0 + 4 * g + i];
                whf[j][i] = cl < 2 ? P[q.wa + cl * AH + col0 + 4 * g + i] : 0.f;           // A[m = cl (action)][k = 4 g + i (unit of the tile)]
            }
        }
        au32x4 wbx[BX ? 2 : 1][BX ? 4 : 1][3];                       // BX: [tile][k-step][h | m | l], eight bf16 per lane
        if constexpr (BX) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 a = wfr[j][2 * t], b = wfr[j][2 * t + 1];
                    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                    split3_pair(a.x, a.y, h0, m0, l0); split3_pair(a.z, a.w, h1, m1, l1);
                    split3_pair(b.x, b.y, h2, m2, l2); split3_pair(b.z, b.w, h3, m3, l3);
                    wbx[j][t][0] = (au32x4){h0, h1, h2, h3}; wbx[j][t][1] = (au32x4){m0, m1, m2, m3}; wbx[j][t][2] = (au32x4){l0, l1, l2, l3};
                }
        }
        int k = 0;
        for (; k < n_steps; ++k) {
            lds_barrier();                                                                         // #1
            if (s_abort) break;
            const bool stamp = dbg_on && wave == 0 && k == dbg_k;
            if (stamp) dbg[8] = clock64();
            float xn = s_raw[cl][g];
            if (use_norm) xn = fminf(fmaxf((xn - s_norm[g]) * s_norm[4 + g], -obs_range), obs_range);
            // Program order = issue order of a wave: a VALU instruction overlaps a running MFMA only when it sits BETWEEN two MFMAs.
            // So: the activations of k-chunks c + 2, c + 3 ride between the branch-layer MFMAs of chunks c, c + 1; tile 0 finishes
            // first and its epilogue (activation + head MFMAs) rides between tile 1's MFMAs; only tile 1's epilogue is exposed.
            // (sched_group_barrier pins the pattern: 1 MFMA, then up to 3 VALU.)
#define XRL_INTERLEAVE(n_mfma)                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < (n_mfma); ++s_) {                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                             \
    }
            f32x4 hf[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) hf[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[c], xn, b0f[c], 0, 0, 0);   // (bias as the C operand)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) hf[c][i] = act_apply_c<ACT>(hf[c][i]);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // all eight first-layer MFMAs in flight before the first activation waits
            __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2][2], lg[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; lg[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if constexpr (BX) {
                // every activation first (the first layer's remaining chunks), then per k-step: split the lane's eight activations, six
                // part products per tile (smallest first), the two tiles' chains alternating
#pragma unroll
                for (int c = 2; c < 8; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hf[c][i] = act_apply_c<ACT>(hf[c][i]);
                acc[0][0] = bmf[0]; acc[1][0] = bmf[1];                                             // (bias as the C operand)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                    split3_pair(hf[2 * t][0], hf[2 * t][1], h0, m0, l0); split3_pair(hf[2 * t][2], hf[2 * t][3], h1, m1, l1);
                    split3_pair(hf[2 * t + 1][0], hf[2 * t + 1][1], h2, m2, l2); split3_pair(hf[2 * t + 1][2], hf[2 * t + 1][3], h3, m3, l3);
                    const au32x4 xh = {h0, h1, h2, h3}, xm = {m0, m1, m2, m3}, xl = {l0, l1, l2, l3};
                    acc[0][0] = AMFMA16B(wbx[0][t][0], xl, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][0], xl, acc[1][0]);
                    acc[0][0] = AMFMA16B(wbx[0][t][2], xh, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][2], xh, acc[1][0]);
                    acc[0][0] = AMFMA16B(wbx[0][t][1], xm, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][1], xm, acc[1][0]);
                    acc[0][0] = AMFMA16B(wbx[0][t][0], xm, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][0], xm, acc[1][0]);
                    acc[0][0] = AMFMA16B(wbx[0][t][1], xh, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][1], xh, acc[1][0]);
                    acc[0][0] = AMFMA16B(wbx[0][t][0], xh, acc[0][0]); acc[1][0] = AMFMA16B(wbx[1][t][0], xh, acc[1][0]);
                }
                if (stamp) dbg[9] = clock64();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h = act_apply_c<ACT>(acc[j][0][i]);
                        MFMA16(whf[j][i], h, lg[j]);
                    }
            } else {
            // ---- tile 0 (+ the first layer's remaining activations)
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                if (c == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[0][0].x, hf[0][0], bmf[0], 0, 0, 0);     // (bias as the C operand)
                else MFMA16(wfr[0][c].x, hf[c][0], acc[0][0]);
                MFMA16(wfr[0][c + 1].x, hf[c + 1][0], acc[0][1]);
                MFMA16(wfr[0][c].y, hf[c][1], acc[0][0]); MFMA16(wfr[0][c + 1].y, hf[c + 1][1], acc[0][1]);
                MFMA16(wfr[0][c].z, hf[c][2], acc[0][0]); MFMA16(wfr[0][c + 1].z, hf[c + 1][2], acc[0][1]);
                MFMA16(wfr[0][c].w, hf[c][3], acc[0][0]); MFMA16(wfr[0][c + 1].w, hf[c + 1][3], acc[0][1]);
                if (c + 2 < 8) {
#pragma unroll
                    for (int cc = c + 2; cc < c + 4; ++cc)
#pragma unroll
                        for (int i = 0; i < 4; ++i) hf[cc][i] = act_apply_c<ACT>(hf[cc][i]);
                }
                XRL_INTERLEAVE(8)
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- tile 1 (+ tile 0's epilogue: activation in place, then its 16 units' share of both logits on the matrix cores)
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                if (c == 0) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[1][0].x, hf[0][0], bmf[1], 0, 0, 0);
                else MFMA16(wfr[1][c].x, hf[c][0], acc[1][0]);
                MFMA16(wfr[1][c + 1].x, hf[c + 1][0], acc[1][1]);
                MFMA16(wfr[1][c].y, hf[c][1], acc[1][0]); MFMA16(wfr[1][c + 1].y, hf[c + 1][1], acc[1][1]);
                if (c < 4) {
#pragma unroll
                    for (int i = 2 * (c / 2); i < 2 * (c / 2) + 2; ++i) {                           // c = 0: units i = 0, 1; c = 2: i = 2, 3
                        const float h = act_apply_c<ACT>(acc[0][0][i] + acc[0][1][i]);
                        MFMA16(whf[0][i], h, lg[0]);
                    }
                }
                MFMA16(wfr[1][c].z, hf[c][2], acc[1][0]); MFMA16(wfr[1][c + 1].z, hf[c + 1][2], acc[1][1]);
                MFMA16(wfr[1][c].w, hf[c][3], acc[1][0]); MFMA16(wfr[1][c + 1].w, hf[c + 1][3], acc[1][1]);
                XRL_INTERLEAVE(10)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (stamp) dbg[9] = clock64();
            // ---- tile 1's epilogue
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h = act_apply_c<ACT>(acc[1][0][i] + acc[1][1][i]);
                MFMA16(whf[1][i], h, lg[1]);
            }
            }
#undef XRL_INTERLEAVE
            // lg[.][i] of lane (g, cl) = logit of action 4 g + i, row cl: actions 0, 1 live in DPP row 0
            if (g == 0) *reinterpret_cast<float2*>(&plog[wave][cl][0]) = make_float2(lg[0][0] + lg[1][0], lg[0][1] + lg[1][1]);
            if (stamp) dbg[10] = clock64();
            lds_barrier();                                                                         // #3
            lds_barrier();                                                                         // #0
        }
    } else if (wave == 5) {
        // ================================================================ envs.step for both actions (lanes 0..31: DPP row 0 action
        // 0, row 1 action 1; the simulator state lives in these lanes' registers); memory.observations[t] (lanes 32..47)
        const int row = cl, e = e0 + row;
        float* po = q.f_obs + ((size_t)q.t0 * n + e) * 4;
        double cps[4] = {0.0, 0.0, 0.0, 0.0};
        if (lane < 32 && e < n) {
            cps[0] = q.cp_state[(size_t)e * 4 + 0]; cps[1] = q.cp_state[(size_t)e * 4 + 1];
            cps[2] = q.cp_state[(size_t)e * 4 + 2]; cps[3] = q.cp_state[(size_t)e * 4 + 3];
        }
        // the state the drawn action of step k (or its auto-reset) left behind: s_sel by the chain wave
        auto adopt = [&](int k) {
            const int sel = s_sel[row];
            const double* src = sel == 2 ? &rs_state[k & 1][row][0] : &ph_state[sel & 1][row][0];
            const double2 s01 = *reinterpret_cast<const double2*>(src), s23 = *reinterpret_cast<const double2*>(src + 2);
            cps[0] = s01.x; cps[1] = s01.y; cps[2] = s23.x; cps[3] = s23.y;
        };
        const size_t tape_row0 = TAPE ? (size_t)(*q.tape_pos) + (size_t)q.t0 : 0;
        auto physics = [&](int k) {                              // outcomes of step k
            if (lane < 32) {
                if constexpr (TAPE) {                            // what the recorded simulator returned (the same for both actions)
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    int fl = 0;
                    const size_t r = tape_row0 + (size_t)k;
                    if (e < n && r < (size_t)q.tape_rows) {
                        o = *reinterpret_cast<const float4*>(q.tape_next_obs + (r * n + e) * 4);
                        fl = (q.tape_term[r * n + e] > 0.f ? 1 : 0) | (q.tape_trunc[r * n + e] > 0.f ? 2 : 0);
                    }
                    *reinterpret_cast<double2*>(&ph_state[g][row][0]) = make_double2((double)o.x, (double)o.y);
                    *reinterpret_cast<double2*>(&ph_state[g][row][2]) = make_double2((double)o.z, (double)o.w);
                    *reinterpret_cast<float4*>(&ph_f[k & 1][g][row][0]) = o;
                    ph_term[k & 1][g][row] = fl;                 // bit 0 terminated, bit 1 truncated (the chain wave reads both from here)
                } else {
                    double x, xd, th, thd;
                    bool term;
                    cartpole_advance(cps, g, x, xd, th, thd, term);
                    *reinterpret_cast<double2*>(&ph_state[g][row][0]) = make_double2(x, xd);
                    *reinterpret_cast<double2*>(&ph_state[g][row][2]) = make_double2(th, thd);
                    *reinterpret_cast<float4*>(&ph_f[k & 1][g][row][0]) = make_float4((float)x, (float)xd, (float)th, (float)thd);
                    ph_term[k & 1][g][row] = term ? 1 : 0;
                }
            }
        };
        physics(0);
        int k = 0;
        for (; k < n_steps; ++k) {
            lds_barrier();                                                                         // #1
            if (s_abort) break;
            if (lane >= 32 && lane < 48) {                       // the normalised observation of the row -> buffer slot t
                const float4 x = normalise(*reinterpret_cast<const float4*>(&s_raw[row][0]));
                if (e < n) *reinterpret_cast<float4*>(po) = x;
            }
            po += (size_t)n * 4;
            lds_barrier();                                                                         // #3
            lds_barrier();                                                                         // #0
            const bool stamp = dbg_on && k == dbg_k;
            if (stamp) dbg[11] = clock64();
            if (lane < 32) adopt(k);
            if (k + 1 < n_steps) physics(k + 1);
            if (stamp) dbg[12] = clock64();
        }
        if (k == n_steps && lane < 16 && e < n) {                // hand the simulator state back
            q.cp_state[(size_t)e * 4 + 0] = cps[0]; q.cp_state[(size_t)e * 4 + 1] = cps[1];
            q.cp_state[(size_t)e * 4 + 2] = cps[2]; q.cp_state[(size_t)e * 4 + 3] = cps[3];
        }
    } else if (wave == 6) {
        // ================================================================ state after an auto-reset into the next episode
        const int row = cl, e = e0 + row;
        const uint64_t env_seed = q.env_seed;
        const size_t tape_row0 = TAPE ? (size_t)(*q.tape_pos) + (size_t)q.t0 : 0;
        auto draw = [&](int k, int ep) {                         // for step k: ep = episode counter after step k - 1
            if constexpr (TAPE) {                                // infos[i]["reset_obs"] of the recorded step
                if (lane < 16) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    const size_t r = tape_row0 + (size_t)k;
                    if (e < n && r < (size_t)q.tape_rows) o = *reinterpret_cast<const float4*>(q.tape_reset_obs + (r * n + e) * 4);
                    *reinterpret_cast<double2*>(&rs_state[k & 1][row][0]) = make_double2((double)o.x, (double)o.y);
                    *reinterpret_cast<double2*>(&rs_state[k & 1][row][2]) = make_double2((double)o.z, (double)o.w);
                    *reinterpret_cast<float4*>(&rs_f[k & 1][row][0]) = o;
                }
                return;
            }
            if (lane < 32) {
                uint32_t o[4], qq[4];
                philox4x32(env_seed, (uint32_t)e, (uint32_t)(ep + 1), g ? STREAM_RESET_B : STREAM_RESET_A, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) qq[j] = __shfl_xor(o[j], 16, 64);
                if (g == 0) {
                    double r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = -0.05 + 0.1 * u01d(o[j], qq[j]);
                    *reinterpret_cast<double2*>(&rs_state[k & 1][row][0]) = make_double2(r[0], r[1]);
                    *reinterpret_cast<double2*>(&rs_state[k & 1][row][2]) = make_double2(r[2], r[3]);
                    *reinterpret_cast<float4*>(&rs_f[k & 1][row][0]) = make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
                }
            }
        };
        draw(0, e < n ? q.cp_episodes[e] : 0);
        int k = 0;
        for (; k < n_steps; ++k) {
            lds_barrier();                                                                         // #1
            if (s_abort) break;
            lds_barrier();                                                                         // #3
            lds_barrier();                                                                         // #0
            if (k + 1 < n_steps) draw(k + 1, ep_lds[row]);
        }
    } else if (wave == 7) {
        // ================================================================ records wave: the step's transition to the buffer, the
        // next raw rows, the sampling uniforms
        const int row = cl, e = e0 + row;
        const bool row_ok = e < n && lane < 16;
        const uint64_t seed = q.seed;
        const uint32_t step0 = q.step + (q.step_dev ? *q.step_dev : 0u) + (uint32_t)q.t0;
        const int T = q.T, t0 = q.t0;
        unsigned* done = q.xchg + XW_DONE + wg;
        const int n4 = (n + 3) & ~3;
        const size_t o0 = (size_t)t0 * n + e;
        float* p_act = q.f_act + o0; float* p_logp = q.f_logp + o0; float* p_term = q.f_term + o0;
        uint8_t* p_seg = q.f_seg + o0; float* p_xn = q.xnext + o0 * 4;
        float* p_rfin = q.ret_final + (size_t)t0 * n4 + e; uint8_t* p_end = q.ended + (size_t)t0 * n4 + e;
        auto draw = [&](int k) {
            if (lane < 16) {
                if (TAPE && q.tape_u) { s_u[row] = e < n ? q.tape_u[(size_t)(t0 + k) * n + e] : 0.f; return; }   // supplied uniforms
                uint32_t rr4[4];
                philox4x32(seed, (uint32_t)e, step0 + (uint32_t)k, STREAM_ACTION, rr4);
                s_u[row] = u01(rr4[0]);
            }
        };
        if (lane < 16) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok) o = *reinterpret_cast<const float4*>(q.obs_raw + (size_t)e * 4);
            *reinterpret_cast<float4*>(&s_raw[row][0]) = o;
        }
        draw(0);
        int k = 0;
        for (; k < n_steps; ++k) {
            lds_barrier();                                                                         // #1
            if (s_abort) break;
            // idle until the env's fate is known: the stores of the previous step have reached L2 by now -- progress word for the
            // trailing readers (steps < k complete)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && k > 0) a_st_dev(done, (unsigned)k);
            // this step's statistics, into registers (the chain wave writes the next step's behind barrier #0)
            const float4 nm = *reinterpret_cast<const float4*>(&s_norm[0]), ni = *reinterpret_cast<const float4*>(&s_norm[4]);
            lds_barrier();                                                                         // #3
            lds_barrier();                                                                         // #0
            if (lane < 16) {
                const float4 rec = *reinterpret_cast<const float4*>(&s_rec[row][0]);
                const int a = (int)rec.x, flags = __float_as_int(rec.z);
                const bool term = (flags & 1) != 0, fin = (flags & 2) != 0;
                const float4 nobs = *reinterpret_cast<const float4*>(&ph_f[k & 1][a][row][0]);
                // (both rows are read and the VALUES selected: `fin ? *lds_ptr : nobs` became a select between an LDS address and the address of
                //  a stack copy of nobs -- a scratch store and a FLAT load through a generic pointer on every step of the chain)
                const float4 rsv = *reinterpret_cast<const float4*>(&rs_f[k & 1][row][0]);
                const float4 robs = make_float4(fin ? rsv.x : nobs.x, fin ? rsv.y : nobs.y, fin ? rsv.z : nobs.z, fin ? rsv.w : nobs.w);
                // get_terminated_values' input: next_obs (before the reset), normalised with this step's statistics (read after the launch)
                float4 nv = nobs;
                if (use_norm) {
                    nv.x = fminf(fmaxf((nobs.x - nm.x) * ni.x, -obs_range), obs_range);
                    nv.y = fminf(fmaxf((nobs.y - nm.y) * ni.y, -obs_range), obs_range);
                    nv.z = fminf(fmaxf((nobs.z - nm.z) * ni.z, -obs_range), obs_range);
                    nv.w = fminf(fmaxf((nobs.w - nm.w) * ni.w, -obs_range), obs_range);
                }
                *reinterpret_cast<float4*>(&s_raw[row][0]) = robs;
                if (row_ok) {
                    *p_act = rec.x;
                    *p_logp = rec.y;
                    *p_term = term ? 1.f : 0.f;
                    *p_seg = (fin || t0 + k == T - 1) ? (uint8_t)(1 | (term ? 6 : 0)) : (uint8_t)0;
                    *reinterpret_cast<float4*>(p_xn) = nv;
                    // what the bookkeeper reads inside the launch: always device scope (its placement is not part of the check)
                    if (fin) a_st_dev(p_rfin, rec.w);
                    a_st_dev(p_end, (uint8_t)(fin ? 1 : 0));
                }
            }
            p_act += n; p_logp += n; p_term += n; p_seg += n; p_xn += (size_t)n * 4; p_rfin += n4; p_end += n4;
            if (k + 1 < n_steps) draw(k + 1);
        }
        if (k == n_steps && row_ok) *reinterpret_cast<float4*>(q.obs_raw + (size_t)e * 4) = *reinterpret_cast<const float4*>(&s_raw[row][0]);
    } else {
        // ================================================================ wave 4, the chain wave: lane (d = DPP row, row) --
        // every lane of a row carries the row's counters, DPP row d reduces dimension d of the observations
        const int row = cl, d = g, e = e0 + row;
        const bool row_ok = e < n;
        const bool single = n_act == 1;                          // one workgroup: the partial sums ARE the batch sums
        unsigned long long* xu = reinterpret_cast<unsigned long long*>(q.xchg);
        const int max_steps = q.max_steps;
        const float gamma = q.gamma;
        const float bh0 = P[q.ba], bh1 = P[q.ba + 1];
        int cp_steps = 0, cp_ep = 0;
        float cp_score = 0.f, rtrack = 0.f;
        float st_mean = 0.f, st_var = 1.f;
        double st_cnt = 0.0;
        // episode statistics of this launch (added to the env's counters once, at the end; integer-valued: exact in any order)
        int ep_cnt = 0, ep_steps = 0;
        double ep_score = 0.0;
        if (row_ok) { cp_steps = q.cp_steps[e]; cp_score = q.cp_score[e]; rtrack = q.ret_track[e]; cp_ep = q.cp_episodes[e]; }
        if (use_norm) { st_mean = q.obs_stats[d]; st_var = q.obs_stats[4 + d]; st_cnt = *q.obs_count; }
        if (lane == 0) { s_abort = 0; s_multi = (q.flags & 1) ? 1 : 0; }
        if (d == 0) ep_lds[row] = cp_ep;
        // message of a step: unit u = 8 * half + j (j = d: sum, 4 + d: sum of squares) of this workgroup -- positions 0..3 of DPP
        // row d hold (lo s1, lo s2, hi s1, hi s2)
        const int unit = (8 * (row >> 1) + 4 * (row & 1) + d) * 16 + wg;
        auto message = [&](double s1, double s2, unsigned tag) -> unsigned long long {
            const double sv = (row & 1) ? s2 : s1;
            const unsigned word = (row & 2) ? (unsigned)__double2hiint(sv) : (unsigned)__double2loint(sv);
            return ((unsigned long long)word << 32) | (unsigned long long)tag;
        };
        // partial sums of the raw observations of this workgroup's rows, dimension d: (sum, sum of squares) in every lane of row d
        double ps1 = 0.0, ps2 = 0.0;
        if (use_norm) {
            const double v = row_ok ? (double)q.obs_raw[(size_t)e * 4 + d] : 0.0;
            ps1 = row16_sum(v); ps2 = row16_sum(v * v);
            if (!single) {
                // placement: every workgroup ORs its XCC id into the launch's mask BEFORE its first message; whoever has seen all
                // first messages sees the complete mask
                if (lane == 0) {
                    unsigned xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    if (wg == 0) __hip_atomic_store(q.status + 1, (int)(xcc & 0xf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(q.status + 2, 1 << (xcc & 0xf));
                    const unsigned seen = atomicOr(q.xchg + XW_MASK, 1u << (xcc & 0xf));
                    asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
                }
                if (row < 4) a_st_dev(xu + 256 + unit, message(ps1, ps2, 1u));   // first message (tag 1, slot 1): always device scope
            }
        }
        bool multi = true;                                       // until the placement is known (first poll): device-scope stores
        int k = 0;
        for (; k < n_steps; ++k) {
            const bool stamp = dbg_on && k == dbg_k;
            long long ts0 = 0, ts1 = 0, ts2 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
            if (stamp) ts0 = clock64();
            // ---------------- (1) statistics of the step: the partial sums of all workgroups
            if (use_norm) {
                double S1 = ps1, S2 = ps2;
                // what of the merge does not depend on the batch sums, ahead of the wait for them
                const double cnt = st_cnt, tot = cnt + (double)n;
                const float nf = (float)n, cntf = (float)cnt, rt = 1.0f / (float)tot;
                if (!single) {
                    const unsigned tag = (unsigned)(k + 1);
                    int spins = 0;
                    const bool live = cl < n_act;
                    const unsigned long long* xs = xu + (tag & 1u) * 256 + lane;
                    unsigned long long u0, u1, u2, u3;
                    for (;;) {
                        // device-scope loads: never served by this CU's L1 (workgroup-scope loads, sc0, are: tried, the poll then spins
                        // on a stale line for ever)
                        u0 = a_ld_dev(xs); u1 = a_ld_dev(xs + 64); u2 = a_ld_dev(xs + 128); u3 = a_ld_dev(xs + 192);
                        const bool ok = !live || ((unsigned)u0 == tag && (unsigned)u1 == tag && (unsigned)u2 == tag && (unsigned)u3 == tag);
                        if (__ballot(!ok) == 0ull) break;
                        if ((++spins & 255) == 0 && (spins > 2000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            if (lane == 0) { __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
                            break;
                        }
                    }
                    // lane (DPP row d, workgroup cl): loads 0 / 2 = lo / hi of that workgroup's sum of dimension d, loads 1 / 3 of its squares
                    const double v1 = live ? __hiloint2double((int)(u2 >> 32), (int)(u0 >> 32)) : 0.0;
                    const double v2 = live ? __hiloint2double((int)(u3 >> 32), (int)(u1 >> 32)) : 0.0;
                    S1 = row16_sum(v1); S2 = row16_sum(v2);
                    if (k == 0) {
                        const unsigned mask = a_ld_dev(q.xchg + XW_MASK);
                        multi = __popc(mask) != 1 || (q.flags & 1);
                        if (lane == 0 && multi && wg == 0) atomicAdd(q.status + 3, 1);
                    }
                }
                if (stamp) ts1 = clock64();
                // RunningMeanStd.update (statistic_tools.py:117-185: batch mean, np.std ** 2, update_from_moments) for dimension d.
                // On the step chain, so: one division (1 / tot, above) shared by the three quotients of update_from_moments, the batch
                // variance straight from the float64 sums (no detour over its root), and 1 / (std + 1e-8) from the hardware's root and
                // reciprocal (1 ulp each): everything within ~2 ulp of the reference's own float32 evaluation.
                {
                    const bool pow2 = (n & (n - 1)) == 0;        // scaling by 1/n is then the exact same number as the division
                    const double inv_n = 1.0 / (double)n;
                    const double m = pow2 ? S1 * inv_n : S1 / n;
                    const float bmean = (float)m;
                    const float bv = (float)fmax((pow2 ? S2 * inv_n : S2 / n) - m * m, 0.0);
                    const float delta = bmean - st_mean;
                    const float new_mean = st_mean + delta * nf * rt;
                    const float M2 = st_var * cntf + bv * nf + (delta * delta) * cntf * nf * rt;
                    const float new_var = M2 * rt;
                    st_mean = new_mean; st_var = new_var; st_cnt = tot;
                    if (row == 0) { s_norm[d] = new_mean; s_norm[4 + d] = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(new_var) + 1e-8f); }
                }
            }
            if (stamp) ts2 = clock64();
            lds_barrier();                                                                         // #1 statistics / raw rows ready
            if (s_abort) break;
            // everything of the tail that does not need the logits
            const float u = s_u[row];
            const float oa = ph_f[k & 1][0][row][d], ob = ph_f[k & 1][1][row][d], orst = rs_f[k & 1][row][d];   // dimension d of the three possible next observations
            const int term_a = ph_term[k & 1][0][row], term_b = ph_term[k & 1][1][row];
            lds_barrier();                                                                         // #3 logits ready
            // ---------------- (4) sample, pick the transition, publish the new partial sums
            if (stamp) ts4 = clock64();
            const float2 p0 = *reinterpret_cast<const float2*>(&plog[0][row][0]), p1 = *reinterpret_cast<const float2*>(&plog[1][row][0]);
            const float2 p2 = *reinterpret_cast<const float2*>(&plog[2][row][0]), p3 = *reinterpret_cast<const float2*>(&plog[3][row][0]);
            const float l0 = ((p0.x + p1.x) + (p2.x + p3.x)) + bh0, l1 = ((p0.y + p1.y) + (p2.y + p3.y)) + bh1;
            // ---- get_actions: Categorical(logits).sample() by inverse CDF on the step's uniform, log-prob.  Two actions: the larger
            // logit's exp(l - max) is exactly 1; hardware exp2 / log2 (1 ulp) for the other terms
            int a;
            float logp;
            {
                const float mx = fmaxf(l0, l1), mn = fminf(l0, l1);
                const float se = 1.0f + __expf(mn - mx);
                const float lse = mx + __logf(se);
                const float c = __expf(l0 - lse);
                a = c > u ? 0 : 1;
                logp = (a ? l1 : l0) - lse;
            }
            // ---- envs.step(acts): the pre-computed transition of the drawn action + auto-reset
            const int tfl = a ? term_b : term_a;
            const bool term = TAPE ? (tfl & 1) != 0 : tfl != 0;
            const int steps = cp_steps + 1;
            const bool trunc = TAPE ? (tfl & 2) != 0 : steps >= max_steps;        // (tape: the recorded simulator's own time limit)
            const bool fin = term || trunc;
            const float score = cp_score + 1.0f;
            const float tr = gamma * rtrack + 1.0f;               // self.returns = gamma * self.returns + rewards (reward 1)
            if (fin) cp_ep += 1;
            if (d == 0) {
                s_sel[row] = fin ? 2 : a; ep_lds[row] = cp_ep;
                *reinterpret_cast<float4*>(&s_rec[row][0]) = make_float4((float)a, logp, __int_as_float((term ? 1 : 0) | (fin ? 2 : 0)), tr);
            }
            if (stamp) ts5 = clock64();
            lds_barrier();                                                                         // #0 the env's fate: records, next step's physics / draws start
            // ---- partial sums of the new raw observations (dimension d of this lane's DPP row) and the message of step k + 1
            if (use_norm && k + 1 < n_steps) {
                const float rv = fin ? orst : (a ? ob : oa);
                const double v = row_ok ? (double)rv : 0.0;
                ps1 = row16_sum(v); ps2 = row16_sum(v * v);
                if (!single && row < 4) a_st(xu + (k & 1) * 256 + unit, message(ps1, ps2, (unsigned)(k + 2)), multi);
            }
            if (stamp) ts6 = clock64();
            if (fin && row_ok) { ep_cnt += 1; ep_steps += steps; ep_score += (double)score; }
            cp_steps = fin ? 0 : steps; cp_score = fin ? 0.f : score; rtrack = fin ? 0.f : tr;
            if (stamp) { dbg[0] = ts0; dbg[1] = ts1; dbg[2] = ts2; dbg[4] = ts4; dbg[5] = ts5; dbg[6] = ts6; dbg[15] = 8; }
        }
        // ---- hand the state back
        if (k == n_steps) {
            if (d == 0 && row_ok) {
                q.cp_steps[e] = cp_steps; q.cp_score[e] = cp_score; q.ret_track[e] = rtrack; q.cp_episodes[e] = cp_ep;
                if (ep_cnt) { atomicAdd(&q.cp_stats[0], (double)ep_cnt); atomicAdd(&q.cp_stats[1], ep_score); atomicAdd(&q.cp_stats[2], (double)ep_steps); }
            }
            if (wg == 0 && use_norm && row == 0) {
                q.obs_stats[d] = st_mean; q.obs_stats[4 + d] = st_var;
                if (d == 0) *q.obs_count = st_cnt;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && !s_abort) a_st_dev(q.xchg + XW_DONE + wg, (unsigned)n_steps);
}

// ---------------------------------------------------------------------------------------------------------------------
// V(obs) for rows [t0 n, (t0 + n_steps) n) of the buffer and V(next_obs) for the rows of that range where a path was cut
// without termination (seg == 1: truncation or buffer full; finish_path(vals[i], i), ppo_agent.py:130-135,153-157).
// Jobs are 32-row tiles; workgroup w takes jobs w, w + G, ...: the value tiles first, then those of its bootstrap tiles that
// hold at least one such row.  Waves 0-3: branch layer of the critic half on the matrix cores (two 16-row tiles per wave
// pass, weights in registers) + the value head's partial dot products; waves 4-7: rows + first layer of the NEXT job and
// the finished values of the previous one.
template <int ACT>
__global__ void __launch_bounds__(ATH) critic_values_kernel(xrl_rollout_run_t q) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float h1[2][32 * ALD];
    __shared__ __attribute__((aligned(16))) float pv[2][32][4];          // partial values [row][matrix wave]
    __shared__ unsigned long long s_need;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = q.n, G = gridDim.x, w = blockIdx.x;
    const float* P = q.params;
    const size_t r0 = (size_t)q.t0 * n, R = (size_t)q.n_steps * n;
    const int n_tiles = (int)((R + 31) / 32);
    const int mine = w < n_tiles ? (n_tiles - 1 - w) / G + 1 : 0;        // tiles w, w + G, ... of either kind
    const int cl = lane & 15, g = lane >> 4;
    // ---- which of this workgroup's bootstrap tiles are needed (one round trip, all candidates at once; <= 64 per workgroup)
    if (wave == 4) {
        bool need = false;
        if (lane < mine && lane < 64) {
            const size_t base = r0 + (size_t)(w + lane * G) * 32;
            for (int i = 0; i < 32; ++i) { const size_t r = base + i; if (r < r0 + R && q.f_seg[r] == 1) need = true; }
        }
        const unsigned long long m = __ballot(need);
        if (lane == 0) s_need = m;
    }
    // ---- weights
    const int vt = tid - 256, vr = vt >> 3, vs = vt & 7;                  // vector threads: row vr, hidden units [16 vs, 16 vs + 16)
    float4 big[16], b0r[4];
    float4 bfr[2][8];
    float bm[2] = {0.f, 0.f}, whr[2] = {0.f, 0.f};
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * wave + 16 * j + cl;
#pragma unroll
            for (int c = 0; c < 8; ++c) bfr[j][c] = *reinterpret_cast<const float4*>(P + q.w1 + (size_t)(AH + col) * AH + 16 * c + 4 * g);
            bm[j] = P[q.b1 + AH + col];
            whr[j] = P[q.wc + col];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) big[j] = *reinterpret_cast<const float4*>(P + q.w0 + (size_t)(vs * 16 + j) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0r[j] = *reinterpret_cast<const float4*>(P + q.b0 + vs * 16 + 4 * j);
    }
    const float bc = P[q.bc];
    __syncthreads();
    const unsigned long long need = s_need;
    const int n_jobs = mine + __popcll(need);
    // job i < mine: value tile w + i G; else the (i - mine)-th set bit of `need`: bootstrap tile
    auto job_tile = [&](int i, bool& boot) -> int {
        boot = i >= mine;
        if (!boot) return w + i * G;
        unsigned long long m = need;
        for (int s = 0; s < i - mine; ++s) m &= m - 1;
        return w + (__ffsll((long long)m) - 1) * G;
    };
    auto first_layer = [&](int i) {                                       // vector waves: rows of job i -> h1[i & 1]
        bool boot;
        const int tile = job_tile(i, boot);
        const size_t r = r0 + (size_t)tile * 32 + vr;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < r0 + R) x = *reinterpret_cast<const float4*>((boot ? q.xnext : q.f_obs) + r * 4);
        float* dst = h1[i & 1] + vr * ALD + vs * 16;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float bq[4] = {b0r[gq].x, b0r[gq].y, b0r[gq].z, b0r[gq].w};
            float o[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 ww = big[gq * 4 + jj];
                float acc = __fmaf_rn(x.x, ww.x, 0.f);
                acc = __fmaf_rn(x.y, ww.y, acc);
                acc = __fmaf_rn(x.z, ww.z, acc);
                acc = __fmaf_rn(x.w, ww.w, acc);
                o[jj] = act_apply_c<ACT>(acc + bq[jj]);
            }
            *reinterpret_cast<float4*>(dst + 4 * gq) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    auto finish = [&](int i) {                                            // wave 4: values of job i from pv[i & 1]
        if (wave != 4 || lane >= 32) return;
        bool boot;
        const int tile = job_tile(i, boot);
        const size_t r = r0 + (size_t)tile * 32 + lane;
        const float4 pp = *reinterpret_cast<const float4*>(&pv[i & 1][lane][0]);
        const float v = ((pp.x + pp.y) + (pp.z + pp.w)) + bc;
        if (r < r0 + R) (boot ? q.bootv : q.f_val)[r] = v;
    };
    if (n_jobs == 0) return;
    if (wave >= 4) first_layer(0);
    lds_barrier();
    for (int i = 0; i < n_jobs; ++i) {
        if (wave < 4) {
            const float* arow = h1[i & 1] + cl * ALD + 4 * g;
            f32x4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[rt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * c), a1 = *reinterpret_cast<const float4*>(arow + 16 * ALD + 16 * c);
                MFMA16(a0.x, bfr[0][c].x, acc[0][0]); MFMA16(a0.x, bfr[1][c].x, acc[0][1]); MFMA16(a1.x, bfr[0][c].x, acc[1][0]); MFMA16(a1.x, bfr[1][c].x, acc[1][1]);
                MFMA16(a0.y, bfr[0][c].y, acc[0][0]); MFMA16(a0.y, bfr[1][c].y, acc[0][1]); MFMA16(a1.y, bfr[0][c].y, acc[1][0]); MFMA16(a1.y, bfr[1][c].y, acc[1][1]);
                MFMA16(a0.z, bfr[0][c].z, acc[0][0]); MFMA16(a0.z, bfr[1][c].z, acc[0][1]); MFMA16(a1.z, bfr[0][c].z, acc[1][0]); MFMA16(a1.z, bfr[1][c].z, acc[1][1]);
                MFMA16(a0.w, bfr[0][c].w, acc[0][0]); MFMA16(a0.w, bfr[1][c].w, acc[0][1]); MFMA16(a1.w, bfr[0][c].w, acc[1][0]); MFMA16(a1.w, bfr[1][c].w, acc[1][1]);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float ha = act_apply_c<ACT>(acc[rt][0][ii] + bm[0]), hb = act_apply_c<ACT>(acc[rt][1][ii] + bm[1]);
                    const float s = row16_sum(ha * whr[0] + hb * whr[1]);
                    if (cl == 0) pv[i & 1][16 * rt + 4 * g + ii][wave] = s;
                }
        } else {
            if (i + 1 < n_jobs) first_layer(i + 1);
            if (i > 0) finish(i - 1);
        }
        lds_barrier();
    }
    finish(n_jobs - 1);
}

__global__ void zero_xchg_kernel(uint32_t* p) {
#pragma unroll
    for (int i = 0; i < XW_WORDS / 512; ++i) p[threadIdx.x + 512 * i] = 0u;
}

static bool g_fast_enabled = true;
bool rollout_fast_enabled() { return g_fast_enabled; }

}  // namespace xrl

using namespace xrl;

namespace xrl { extern bool g_fast_enabled_ppo; }
static int g_rollout_bx = 1;                       // xrl_set_rollout_split_products: the actor rollout's branch layer as exact 3-way bf16 splits
extern "C" int xrl_set_rollout_split_products(int on) {
    g_rollout_bx = on ? 1 : 0;
    return XRL_OK;
}
extern "C" int xrl_set_fast_kernels(int enable) {
    xrl::g_fast_enabled = enable != 0;
    xrl::g_fast_enabled_ppo = enable != 0;
    return XRL_OK;
}

static int check_run(const xrl_rollout_run_t& q, const char* who) {
    if (!(q.params && q.n > 0 && q.T >= 1 && q.t0 >= 0 && q.n_steps >= 1 && q.t0 + q.n_steps <= q.T)) {
        set_error("%s: invalid sizes (n %d, T %d, t0 %d, n_steps %d)", who, q.n, q.T, q.t0, q.n_steps);
        return XRL_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(q.params) & 15) || (q.w0 & 3) || (q.b0 & 3) || (q.w1 & 3)) {
        set_error("%s: parameter blocks must be 16-byte aligned", who);
        return XRL_EINVAL;
    }
    return XRL_OK;
}

extern "C" int xrl_rollout_cartpole_max_envs(void) {
    // the whole-rollout launch keeps its workgroups (AR envs each + the bookkeeper) resident on ONE XCD, one per CU
    const int wg = device_cu_count() / 8 - 1;
    return AR * (wg < AMAXWG ? (wg > 0 ? wg : 0) : AMAXWG);
}

extern "C" int xrl_rollout_cartpole_run(const xrl_rollout_run_t* qq, xrl_stream_t stream) {
    XRL_CHECK_ARG(qq != nullptr);
    const xrl_rollout_run_t& q = *qq;
    if (int rc = check_run(q, "xrl_rollout_cartpole_run")) return rc;
    XRL_CHECK_ARG(q.n <= AR * AMAXWG);
    XRL_CHECK_ARG(q.obs_raw && q.obs_stats && q.obs_count && q.ret_stats && q.ret_count && q.ret_track);
    XRL_CHECK_ARG(q.cp_state && q.cp_steps && q.cp_episodes && q.cp_score && q.cp_stats);
    XRL_CHECK_ARG(q.f_obs && q.f_act && q.f_logp && q.f_rew && q.f_term && q.f_seg && q.xnext && q.ended && q.ret_final);
    XRL_CHECK_ARG(q.xchg && q.status);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(q.ended) & 3) == 0 && (reinterpret_cast<uintptr_t>(q.xchg) & 7) == 0);
    const int n_wg = (q.n + AR - 1) / AR + 1;                           // actors + the bookkeeper
    XRL_CHECK_ARG(n_wg <= device_cu_count() / 8);                       // all resident on ONE XCD, one per CU
    // zeroed by a kernel, not by hipMemsetAsync: a memset node of a captured graph does not order the kernel nodes around it
    // (ROCm 7.2, tools/stress_determinism.py)
    hipLaunchKernelGGL(zero_xchg_kernel, dim3(1), dim3(512), 0, as_stream(stream), q.xchg);
    if (q.tape_next_obs) {
        XRL_CHECK_ARG(q.tape_reset_obs && q.tape_term && q.tape_trunc && q.tape_pos && q.tape_rows >= 1);
        XRL_CHECK_ARG(((reinterpret_cast<uintptr_t>(q.tape_next_obs) | reinterpret_cast<uintptr_t>(q.tape_reset_obs)) & 15) == 0);
        if (g_rollout_bx) { XRL_ACT_DISPATCH(q.act, hipLaunchKernelGGL((actor_rollout_kernel<ACT, true, true>), dim3(8 * n_wg), dim3(ATH), 0, as_stream(stream), q);) }
        else { XRL_ACT_DISPATCH(q.act, hipLaunchKernelGGL((actor_rollout_kernel<ACT, true, false>), dim3(8 * n_wg), dim3(ATH), 0, as_stream(stream), q);) }
    } else {
        XRL_CHECK_ARG(q.tape_u == nullptr);                              // (supplied uniforms ride with a tape only)
        if (g_rollout_bx) { XRL_ACT_DISPATCH(q.act, hipLaunchKernelGGL((actor_rollout_kernel<ACT, false, true>), dim3(8 * n_wg), dim3(ATH), 0, as_stream(stream), q);) }
        else { XRL_ACT_DISPATCH(q.act, hipLaunchKernelGGL((actor_rollout_kernel<ACT, false, false>), dim3(8 * n_wg), dim3(ATH), 0, as_stream(stream), q);) }
    }
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_rollout_cartpole_values(const xrl_rollout_run_t* qq, xrl_stream_t stream) {
    XRL_CHECK_ARG(qq != nullptr);
    const xrl_rollout_run_t& q = *qq;
    if (int rc = check_run(q, "xrl_rollout_cartpole_values")) return rc;
    XRL_CHECK_ARG(q.f_obs && q.xnext && q.f_seg && q.f_val && q.bootv);
    const long long R = (long long)q.n_steps * q.n;
    const int n_tiles = (int)((R + 31) / 32);
    int grid = device_cu_count();
    if (grid <= 0) grid = 256;
    if (grid > n_tiles) grid = n_tiles;
    XRL_CHECK_ARG((n_tiles + grid - 1) / grid <= 64);                   // candidates of one workgroup: one ballot
    XRL_ACT_DISPATCH(q.act, hipLaunchKernelGGL((critic_values_kernel<ACT>), dim3(grid), dim3(ATH), 0, as_stream(stream), q);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

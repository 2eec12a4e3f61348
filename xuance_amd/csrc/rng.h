// Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// key = 64-bit seed, counter = (c0, c1, c2, 0): every (env, step, stream) triple has its own reproducible draw, so
// results do not depend on launch geometry or on hipGraph replay.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace xrl {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ void philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t (&out)[4]) {
    uint32_t c[4] = {c0, c1, c2, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int i = 0; i < 10; ++i) philox_round(c, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }          // [0,1)
__device__ __forceinline__ double u01d(uint32_t a, uint32_t b) {
    return (double)((((uint64_t)a << 21) ^ (uint64_t)b) & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}

// Standard normal for the SYNTHETIC providers' noise (not a reference quantity): Box-Muller on one Philox call with the hardware's
// log2 / sqrt / cos (v_cos_f32 takes revolutions: cos(2 pi u) is one instruction) -- the library's logf / cosf cost ~1.5 k cycles
// per draw on the step chain of the whole-rollout launch (csrc/rollout_wide.hip), these ~100.
__device__ __forceinline__ float provider_normal(uint64_t seed, uint32_t e, uint32_t step, uint32_t stream) {
    uint32_t r[4];
    philox4x32(seed, e, step, stream, r);
    const float u1 = fmaxf(u01(r[0]), 1e-7f), u2 = u01(r[1]);
    return __builtin_amdgcn_sqrtf(-2.f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
}
// Standard normal of the policy's Normal(mu, std).sample() (xrl_policy_sample, xrl_wide_act_step, xrl_rollout_wide_run: one
// definition, the same draw for (seed, env, step, action) on every path): Box-Muller on a Philox call, hardware log2 / sqrt / cos
// (|error| ~ 1e-6 against the float64 evaluation the oracle restates it with; the reference's own draws are torch's generator,
// which no engine can reproduce)
__device__ __forceinline__ float policy_normal(uint64_t seed, uint32_t e, uint32_t step, uint32_t j) {
    uint32_t r[4];
    philox4x32(seed, e, step, 0x47415500u + j, r);            // STREAM_GAUSS + j
    const float u1 = fmaxf(u01(r[0]), 5.96e-8f), u2 = u01(r[1]);
    return __builtin_amdgcn_sqrtf(-2.f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
}
// tanh of the providers' dynamics: 1 - 2 / (exp(2 x) + 1) on the hardware's exp2 / rcp (|error| ~ 1e-7)
__device__ __forceinline__ float provider_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

constexpr uint32_t STREAM_ACTION = 0x41435431u, STREAM_GAUSS = 0x47415500u, STREAM_RESET_A = 0x52455345u,
                   STREAM_RESET_B = 0x52455346u, STREAM_EGREEDY = 0x45475200u;


// One row of xrl_marl_select_actions (off_policy_marl.py:212-255): masked greedy action of q_row unless the step's coin lands
// under epsilon, then the k-th AVAILABLE action, k uniform.  Shared by marl_select_kernel and the one-launch acting step.
// (the choice itself, given the step's coin and the row's uniform: the one-thread-per-output acting kernel draws them in idle lanes
//  while its weights are in flight)
__device__ __forceinline__ int marl_pick_row(const float* q_row, const float* av, int A, float coin, float u, float eps) {
    int best = 0, n_avail = 0;
    float bv = (av && av[0] == 0.f) ? -1e10f : q_row[0];
    for (int j = 0; j < A; ++j) {
        const bool ok = !av || av[j] != 0.f;
        n_avail += ok;
        const float v = ok ? q_row[j] : -1e10f;
        if (j > 0 && v > bv) { bv = v; best = j; }
    }
    int a = best;
    if (coin < eps) {
        const int na = n_avail > 1 ? n_avail : 1;
        int kth = (int)(u * (float)na), seen = 0;
        if (kth > na - 1) kth = na - 1;
        a = 0;
        for (int j = 0; j < A; ++j) {
            const bool ok = !av || av[j] != 0.f;
            if (ok) { if (seen == kth) { a = j; break; } ++seen; }
        }
    }
    return a;
}
__device__ __forceinline__ float marl_step_coin(uint64_t seed, uint32_t step) {      // the step coin: same counter for every row
    uint32_t c[4];
    philox4x32(seed, 0xFFFFFFFFu, step, STREAM_EGREEDY, c);
    return u01(c[0]);
}
__device__ __forceinline__ float marl_row_uniform(uint64_t seed, uint32_t step, int r) {
    uint32_t rr[4];
    philox4x32(seed, (uint32_t)r, step, STREAM_EGREEDY + 1u, rr);
    return u01(rr[0]);
}
__device__ __forceinline__ int marl_select_row(const float* q_row, const float* av, int A, uint64_t seed, uint32_t step, int r,
                                               float eps, const float* coin_in, const float* uniforms) {
    const float coin = coin_in ? *coin_in : marl_step_coin(seed, step);
    float u = 0.f;
    if (coin < eps) u = uniforms ? uniforms[r] : marl_row_uniform(seed, step, r);
    return marl_pick_row(q_row, av, A, coin, u, eps);
}

// Draw of xrl_sample_replay_indices for batch row b (same Philox stream: the fused draw + gather picks the same rows).
struct ReplayDraw {
    const int32_t* size_dev; uint64_t seed; uint32_t counter; const uint32_t* counter_dev; int64_t* idx_out;
};
__device__ __forceinline__ int64_t replay_draw(const ReplayDraw& s, int b, int n_envs, int n_size) {
    const uint32_t ctr = s.counter + (s.counter_dev ? *s.counter_dev : 0u);
    int size = *s.size_dev;
    size = size < 1 ? 1 : (size > n_size ? n_size : size);
    uint32_t r[4];
    philox4x32(s.seed, (uint32_t)b, ctr, 0x53414D50u, r);
    const int env = (int)(((uint64_t)r[0] * (uint64_t)n_envs) >> 32);
    const int step = (int)(((uint64_t)r[1] * (uint64_t)size) >> 32);
    return (int64_t)env * n_size + step;
}


}  // namespace xrl

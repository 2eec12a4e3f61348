// One row of xrl_policy_sample (OnPolicyAgent.get_actions, core/on_policy.py:128-169): shared by policy_sample_kernel (csrc/rollout.hip)
// and the tail launch of the on-policy acting step (csrc/act_tail.hip).
#pragma once
#include "common.h"
#include "rng.h"

namespace xrl {

// the ACTOR half of a row: action from the head outputs h[0..A), its log-prob; what needs no value (csrc/ppo_trunk_bx.hip's forward-only
// instances run the two heads in different workgroups: the critic's writes val_out / bootv_prev itself)
__device__ __forceinline__ void policy_sample_actor(const xrl_sample_t& p, int e, const float* h) {
    const int A = p.A;
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    float logp;
    if (!p.gaussian) {
        float u;
        if (p.noise) u = p.noise[e];
        else { uint32_t r[4]; philox4x32(p.seed, (uint32_t)e, step, STREAM_ACTION, r); u = u01(r[0]); }
        float mx = h[0];
        for (int j = 1; j < A; ++j) mx = fmaxf(mx, h[j]);
        float se = 0.f;
        for (int j = 0; j < A; ++j) se += expf(h[j] - mx);
        const float lse = mx + logf(se);
        // inverse CDF over softmax probabilities accumulated left to right in float32
        int a = A - 1;
        float c = 0.f;
        for (int j = 0; j < A; ++j) {
            c += expf(h[j] - lse);
            if (c > u) { a = j; break; }
        }
        logp = h[a] - lse;                               // Categorical.log_prob (distributions.py:147-148)
        p.act_out[e] = (float)a;
        if (p.env_action) p.env_action[e] = a;
    } else {
        logp = 0.f;
        for (int j = 0; j < A; ++j) {
            float z;
            if (p.noise) z = p.noise[(size_t)e * A + j];
            else {
                z = policy_normal(p.seed, (uint32_t)e, step, (uint32_t)j);
            }
            const float ls = p.log_std[j], sd = expf(ls);
            const float x = h[j] + sd * z;                // Normal(mu, std).sample()
            const float df = x - h[j];
            logp += -(df * df) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;
            p.act_out[(size_t)e * A + j] = x;
            if (p.env_action_f) p.env_action_f[(size_t)e * A + j] = x;
        }
    }
    p.logp_out[e] = logp;
}

// h: the row's head outputs (cols [0, A) actor output, col A value); boot_value: the value of row n + e (read only with p.bootv_prev)
__device__ __forceinline__ void policy_sample_one(const xrl_sample_t& p, int e, const float* h, float boot_value) {
    if (p.bootv_prev) p.bootv_prev[e] = boot_value;
    if (!p.act_out) return;                                // bootstrap-only launch (end of a rollout)
    policy_sample_actor(p, e, h);
    if (p.val_out) p.val_out[e] = h[p.A];                  // actor-only policies (VanillaPolicyGradient) have no value column
}

}  // namespace xrl

// Per-sample PPO-clip arithmetic shared by ppo_loss.hip (layered path) and ppo_fused.hip (fused minibatch kernel).
// Reference: xuance/torch/learners/policy_gradient/ppo_learner.py:52-55,70 with the autograd rules of torch.clamp
// (gradient where lo <= x <= hi) and torch.minimum (0.5/0.5 on ties).
#pragma once
#include <hip/hip_runtime.h>

namespace xrl {

constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

struct Surrogate {
    float ratio, s1, s2, dlogp;
    int clipped;
};

__device__ __forceinline__ Surrogate surrogate(float logp, float old_logp, float adv, float lo, float hi, float invM) {
    Surrogate r;
    r.ratio = expf(logp - old_logp);                        // :52
    const float rc = fminf(fmaxf(r.ratio, lo), hi);
    r.s1 = rc * adv;                                        // :53
    r.s2 = adv * r.ratio;                                   // :54
    const float inside = (r.ratio >= lo && r.ratio <= hi) ? 1.f : 0.f;
    const float w1 = r.s1 < r.s2 ? 1.f : (r.s1 == r.s2 ? 0.5f : 0.f);
    const float dratio = -(w1 * inside * adv + (1.f - w1) * adv) * invM;   // d(-mean(min(s1,s2)))/d ratio
    r.dlogp = dratio * r.ratio;
    r.clipped = (r.ratio < lo) || (r.ratio > hi);           // :70
    return r;
}

// A2C actor term (a2c_learner.py:47): a_loss = -(adv * log_prob).mean().  Reported through the same fields: s1 = s2 =
// adv * log_prob (so that -sum(min(s1, s2))/M is the loss), d a_loss / d log_prob = -adv / M, nothing is ever clipped.
__device__ __forceinline__ Surrogate surrogate_a2c(float logp, float adv, float invM) {
    Surrogate r;
    r.ratio = 1.f;
    r.s1 = r.s2 = adv * logp;
    r.dlogp = -adv * invM;
    r.clipped = 0;
    return r;
}

// PPO-KL actor term (ppokl_learner.py:57-58): -(ratio * adv).mean() (+ kl_coef * kl, added by the caller): no clipping.
__device__ __forceinline__ Surrogate surrogate_kl(float logp, float old_logp, float adv, float invM) {
    Surrogate r;
    r.ratio = expf(logp - old_logp);
    r.s1 = r.s2 = r.ratio * adv;
    r.dlogp = -adv * r.ratio * invM;
    r.clipped = 0;
    return r;
}

}  // namespace xrl

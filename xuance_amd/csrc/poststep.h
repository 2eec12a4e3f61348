// Per-step bookkeeping of PPO_Agent.train (ppo_agent.py:128,144-157) as a device function: one workgroup of NT threads.
// Used by poststep_kernel (rollout.hip, a launch of its own) and by the acting launch of the two-branch Gaussian class
// (ppo_wide.hip), where one extra workgroup does the previous step's bookkeeping beside the acting workgroups.
#pragma once
#include "common.h"

namespace xrl {

template <int NT>
__device__ __forceinline__ void poststep_body(const xrl_poststep_t& p, unsigned long long* ended_mask /* [64] in LDS */) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x, n = p.n;
    // reward normalisation uses the return statistics BEFORE this step's episode-end updates (ppo_agent.py:128)
    float rstd = sqrtf(*p.ret_var);
    rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
    for (int e = tid; e < n; e += NT) {
        const float r = p.reward[e];
        float rn = r;
        if (p.use_rewnorm) rn = fminf(fmaxf(r / rstd, -p.rew_range), p.rew_range);
        p.rew_out[e] = rn;
        if (p.pg_bootv) p.pg_bootv[e] = rn;
        const bool term = p.terminated[e] != 0.f, trunc = p.truncated[e] != 0.f;
        p.term_out[e] = term ? 1.f : 0.f;
        uint8_t sg = 0;
        if (term || trunc || p.last_step) sg = 1 | (term ? 6 : 0);   // finish_path(0.0, i): float64-carry form (2), bootstrap value 0 (4)
        p.seg_out[e] = sg;
        p.ret_track[e] = p.gamma * p.ret_track[e] + r;              // self.returns = gamma * self.returns + rewards
    }
    // normalised next observation with the current observation statistics (get_terminated_values, on_policy.py:109)
    const int total = p.next_obs_norm ? n * p.D : 0;                 // (NULL: the consumer normalises the rows itself)
    for (int i = tid; i < total; i += NT) {
        const int e = i / p.D, d = i - e * p.D;
        float v = p.next_obs[(size_t)e * p.D + d];
        if (p.use_obsnorm) {
            v = (v - p.obs_mean[d]) / (sqrtf(p.obs_var[d]) + 1e-8f);
            v = fminf(fmaxf(v, -p.obs_range), p.obs_range);
        }
        p.next_obs_norm[(size_t)e * p.ld_next + d] = v;
    }
    __syncthreads();
    // ret_rms.update(self.returns[i:i+1]) for every finished env IN ENV ORDER (ppo_agent.py:146-149): sequential
    // single-sample merges, then self.returns[i] = 0.
    for (int base = 0; base < n; base += 4096) {
        const int cnt = min(4096, n - base);
        for (int w = tid >> 6; w * 64 < cnt; w += NT / 64) {
            const int e = base + w * 64 + (tid & 63);
            const bool ended = (e < n) && (p.terminated[e] != 0.f || p.truncated[e] != 0.f);
            const unsigned long long m = __ballot(ended);
            if ((tid & 63) == 0) ended_mask[w] = m;
        }
        __syncthreads();
        if (tid == 0) {
            float mean = *p.ret_mean, var = *p.ret_var;
            double count = *p.ret_count;
            for (int w = 0; w * 64 < cnt; ++w) {
                unsigned long long m = ended_mask[w];
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int e = base + w * 64 + b;
                    const float bm = p.ret_track[e];                 // batch_mean of one sample; batch_var = 0, count 1
                    const double tot = count + 1.0;
                    const float delta = bm - mean;
                    const float new_mean = mean + delta * 1.0f / (float)tot;
                    const float m_a = var * (float)count;
                    const float m_b = 0.f * 1.0f;
                    const float M2 = m_a + m_b + (delta * delta) * (float)count * 1.0f / (float)tot;
                    mean = new_mean; var = M2 / (float)tot; count = tot;
                    p.ret_track[e] = 0.f;
                    if (p.pg_bootv && !p.last_step) {                // the closing value of THIS env's path: its reward over the statistics
                        float v = p.reward[e];                      // as they are now (get_terminated_values is called after the update)
                        if (p.use_rewnorm) {
                            const float sd = fminf(fmaxf(sqrtf(var), 0.1f), 100.f);
                            v = fminf(fmaxf(v / sd, -p.rew_range), p.rew_range);
                        }
                        p.pg_bootv[e] = v;
                    }
                }
            }
            *p.ret_mean = mean; *p.ret_var = var; *p.ret_count = count;
        }
        __syncthreads();
    }
}

}  // namespace xrl

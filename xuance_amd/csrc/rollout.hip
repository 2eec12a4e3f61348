// Rollout-side kernels: running observation statistics + normalisation, action sampling from the policy head,
// device-resident CartPole-v1 physics, and the per-step bookkeeping of the on-policy agent loop.
// Reference arithmetic: xuance/common/statistic_tools.py:117-185 (RunningMeanStd), xuance/torch/agents/base/agent.py:262-294
// (_process_observation/_process_reward), core/on_policy.py:128-169 (get_actions), policy_gradient/ppo_agent.py:111-181
// (train loop), core/off_policy.py:129-148 (epsilon-greedy).  CartPole equations: Barto, Sutton & Anderson (1983) as
// published with Gymnasium's classic_control/cartpole.py (not part of the reference tree).
// All of these are latency-bound at the reference's sizes (n_envs <= a few thousand): each is ONE small launch per
// step with coalesced [env]-major accesses, meant to be replayed from a hipGraph.
#include "common.h"
#include "rng.h"
#include "cartpole.h"
#include "poststep.h"
#include "sample.h"

namespace xrl {

// ------------------------------------------------------------------------------------------------ running mean/std

constexpr int RMS_THREADS = 1024;
constexpr int RMS_MAXD = 64;

constexpr int RMS_STAGE = 16;
// part2[j][d] = sum of part[r][d] over r = j, j + 16, ... < R (threads [0, 16 D)); ends in a barrier
__device__ __forceinline__ void rms_stage_sum(const double* part, double* part2, int D, int R, int tid) {
    if (tid < RMS_STAGE * D) {
        const int d = tid % D, j = tid / D;
        double t = 0.0;
        for (int r = j; r < R; r += RMS_STAGE) t += part[r * D + d];
        part2[j * D + d] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ void rms_normalize_body(const xrl_rms_t& p) {
#pragma clang fp contract(off)
    __shared__ double part[RMS_THREADS];
    __shared__ double part2[RMS_STAGE * RMS_MAXD];
    __shared__ double bmean[RMS_MAXD], bvar[RMS_MAXD];
    __shared__ float s_mean[RMS_MAXD], s_std[RMS_MAXD];
    const int D = p.D, n = p.n, tid = threadIdx.x;
    if (p.update) {
        // batch moments over the env axis (np.mean / np.std, axis=0) accumulated in float64
        const int R = RMS_THREADS / D;                 // row-lanes per dimension
        const int d = tid % D, r0 = tid / D;
        const bool live = r0 < R;
        double s = 0.0;
        if (live) for (int r = r0; r < n; r += R) s += (double)p.x[(size_t)r * p.ld_x + d];
        part[tid] = live ? s : 0.0;
        __syncthreads();
        // (the R partial sums of a dimension meet in two stages -- 16 threads per dimension take every 16th, then one thread the 16 --
        //  instead of one thread walking all R of them: that loop was a chain of ~170 dependent LDS reads, twice per launch, a third of
        //  the launch's 10 us; float64 sums of float32 values: the order does not reach the float32 results)
        rms_stage_sum(part, part2, D, R, tid);
        if (tid < D) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < RMS_STAGE; ++j) t += part2[j * D + tid];
            bmean[tid] = (double)(float)(t / n);       // np.mean returns float32
        }
        __syncthreads();
        double q = 0.0;
        if (live) {
            const double m = bmean[d];
            for (int r = r0; r < n; r += R) { const double df = (double)p.x[(size_t)r * p.ld_x + d] - m; q += df * df; }
        }
        part[tid] = live ? q : 0.0;
        __syncthreads();
        rms_stage_sum(part, part2, D, R, tid);
        if (tid < D) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < RMS_STAGE; ++j) t += part2[j * D + tid];
            const float bstd = (float)sqrt(t / n);     // np.std -> float32
            bvar[tid] = (double)(bstd * bstd);         // batch_var = np.square(batch_std)
        }
        __syncthreads();
        if (tid < D) {
            // update_from_moments (statistic_tools.py:173-185), float32 arrays with Python-float count
            const double cnt = *p.count, tot = cnt + (double)n;
            const float mean = p.mean[tid], var = p.var[tid];
            const float bm = (float)bmean[tid], bv = (float)bvar[tid];
            const float delta = bm - mean;
            const float new_mean = mean + delta * (float)n / (float)tot;
            const float m_a = var * (float)cnt;
            const float m_b = bv * (float)n;
            const float M2 = m_a + m_b + (delta * delta) * (float)cnt * (float)n / (float)tot;
            const float new_var = M2 / (float)tot;
            p.mean[tid] = new_mean; p.var[tid] = new_var;
            s_mean[tid] = new_mean; s_std[tid] = sqrtf(new_var);
        }
        __syncthreads();
        if (tid == 0) *p.count = *p.count + (double)n;
    } else {
        if (tid < D) { s_mean[tid] = p.mean[tid]; s_std[tid] = sqrtf(p.var[tid]); }
        __syncthreads();
    }
    const int total = n * D;
    for (int i = tid; i < total; i += RMS_THREADS) {
        const int r = i / D, d = i - r * D;
        float v = p.x[(size_t)r * p.ld_x + d];
        if (p.normalize) {
            v = (v - s_mean[d]) / (s_std[d] + 1e-8f);
            v = fminf(fmaxf(v, -p.range), p.range);
        }
        if (p.out0) p.out0[(size_t)r * p.ld0 + d] = v;
        if (p.out1) p.out1[(size_t)r * p.ld1 + d] = v;
    }
}

__global__ void __launch_bounds__(RMS_THREADS) rms_normalize_kernel(xrl_rms_t p) { rms_normalize_body(p); }

// ------------------------------------------------------------------------------------------------ policy sampling

__global__ void __launch_bounds__(256) policy_sample_kernel(xrl_sample_t p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    policy_sample_one(p, e, p.heads + (size_t)e * p.ld, p.bootv_prev ? p.heads[(size_t)(p.n + e) * p.ld + p.A] : 0.f);
}

// ------------------------------------------------------------------------------------------------ CartPole-v1

__global__ void __launch_bounds__(256) cartpole_step_kernel(xrl_cartpole_t p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    cartpole_step_one(p, e);
}

__global__ void __launch_bounds__(256) cartpole_reset_kernel(xrl_cartpole_t p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    double* s = p.state + (size_t)e * 4;
    cartpole_reset(s, p.seed, e, 0u);
    p.steps[e] = 0; p.episodes[e] = 0; p.ep_score[e] = 0.f;
    float* o = p.obs + (size_t)e * 4;
    o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3];
}

// ------------------------------------------------------------------------------------------------ synthetic control env
// MuJoCo-shaped input provider (no simulator in this image): state' = tanh(state . A + clip(a) . B)
// + 0.01 N(0,1), reward = state'[0] - 0.1 |a|^2, truncation after max_steps, reset to 0.1 N(0,1); same auto-reset
// contract as the CartPole kernel.  One thread per env; the noise is Philox keyed by (seed, env, step).
__device__ __forceinline__ float synth_normal(uint64_t seed, uint32_t e, uint32_t step, uint32_t j) {
    return provider_normal(seed, e, step, 0x53594E00u + j);
}

// One wavefront per env: lane j < D owns state component j (the first version ran one THREAD per env: a serial 17 x 23
// mat-vec with strided loads, 17 Philox normals and tanhf's per thread, 30 us per step for 128 envs).
__global__ void __launch_bounds__(256) synth_control_kernel(xrl_synth_ctl_t p, int reset) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (e >= p.n) return;
    const int D = p.D, Ad = p.A;
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    float* st = p.state + (size_t)e * D;
    if (reset) {
        if (j < D) { const float v = 0.1f * synth_normal(p.seed, (uint32_t)e, 0xffffff00u, (uint32_t)j); st[j] = v; p.obs[(size_t)e * D + j] = v; }
        if (j == 0) { p.steps[e] = 0; p.ep_score[e] = 0.f; }
        return;
    }
    float pen = 0.f, yj = 0.f;
    {
        float acc = 0.f;
        const int jj = j < D ? j : 0;
        for (int k = 0; k < D; ++k) acc += st[k] * p.Amat[k * D + jj];
        for (int i = 0; i < Ad; ++i) {
            const float ai = fminf(fmaxf(p.action[(size_t)e * Ad + i], -1.f), 1.f);
            pen += ai * ai;
            acc += ai * p.Bmat[i * D + jj];
        }
        yj = provider_tanh(acc) + 0.01f * synth_normal(p.seed, (uint32_t)e, step, (uint32_t)jj);
    }
    const float y0 = __shfl(yj, 0, 64);
    const float rew = y0 - 0.1f * pen;
    const int steps = p.steps[e] + 1;                     // (every lane reads the old value before lane 0 writes the new one)
    const bool trunc = steps >= p.max_steps;
    const float score = p.ep_score[e] + rew;
    if (j < D) {
        p.next_obs[(size_t)e * D + j] = yj;
        const float v = trunc ? 0.1f * synth_normal(p.seed, (uint32_t)e, step, 64u + (uint32_t)j) : yj;
        st[j] = v; p.obs[(size_t)e * D + j] = v;
    }
    if (j == 0) {
        p.reward[e] = rew; p.terminated[e] = 0.f; p.truncated[e] = trunc ? 1.f : 0.f;
        if (trunc) {
            atomicAdd(&p.stats[0], 1.0); atomicAdd(&p.stats[1], (double)score); atomicAdd(&p.stats[2], (double)steps);
            p.steps[e] = 0; p.ep_score[e] = 0.f;
        } else {
            p.steps[e] = steps; p.ep_score[e] = score;
        }
    }
}

// ------------------------------------------------------------------------------------------------ synthetic multi-agent env
// SMAC-3m-shaped input provider (no simulator in this image): N agents, per-agent observations, a global state and
// action-availability masks drawn from Philox streams keyed by (seed, env, step); team reward from the chosen actions and
// the state; episodes end with probability p_term per step or at max_steps; auto-reset like the reference's vector envs
// (the returned next_* are the terminal ones, buf_* already hold the first step of the next episode).  One wavefront per env.
__device__ __forceinline__ float synth_uniform(uint64_t seed, uint32_t e, uint32_t step, uint32_t j) {
    uint32_t r[4];
    philox4x32(seed, e, step, 0x554E4900u + j, r);
    return u01(r[0]);
}

__global__ void __launch_bounds__(256) synth_marl_kernel(xrl_synth_marl_t p, int reset) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= p.n) return;
    const int N = p.N, NO = p.N * p.O, NA = p.N * p.A, S = p.S;
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    // stream ids: 2 * step (+1 for the post-reset draw); element spaces: obs [0, NO), state [4096, ..), avail [8192, ..)
    auto draw = [&](uint32_t st, float* obs, float* state, float* avail) {
        for (int j = lane; j < NO; j += 64) obs[(size_t)e * NO + j] = synth_normal(p.seed, (uint32_t)e, st, (uint32_t)j);
        for (int j = lane; j < S; j += 64) state[(size_t)e * S + j] = synth_normal(p.seed, (uint32_t)e, st, 4096u + (uint32_t)j);
        for (int j = lane; j < NA; j += 64)
            avail[(size_t)e * NA + j] = (j % p.A == 0 || synth_uniform(p.seed, (uint32_t)e, st, 8192u + (uint32_t)j) < 0.7f) ? 1.f : 0.f;
    };
    if (reset) {
        draw(0xfffffff0u, p.buf_obs, p.buf_state, p.buf_avail);
        if (lane == 0) { p.steps[e] = 0; p.done[e] = 0.f; p.end_step[e] = 0; }
        return;
    }
    float asum = 0.f;
    for (int a = 0; a < N; ++a) asum += (float)p.action[(size_t)e * N + a];
    const float* acted_state = p.prev_state ? p.prev_state : p.buf_state;     // two-buffer mode: buf_* are the OTHER buffers
    const float rew = asum / (float)N / (float)p.A + 0.1f * acted_state[(size_t)e * S];
    const int steps = p.steps[e] + 1;
    const bool term = synth_uniform(p.seed, (uint32_t)e, 2u * step, 16384u) < p.p_term;
    const bool trunc = !term && steps >= p.max_steps;
    const bool done = term || trunc;
    draw(2u * step, p.next_obs, p.next_state, p.next_avail);
    if (done) {
        draw(2u * step + 1u, p.buf_obs, p.buf_state, p.buf_avail);
    } else {
        for (int j = lane; j < NO; j += 64) p.buf_obs[(size_t)e * NO + j] = p.next_obs[(size_t)e * NO + j];
        for (int j = lane; j < S; j += 64) p.buf_state[(size_t)e * S + j] = p.next_state[(size_t)e * S + j];
        for (int j = lane; j < NA; j += 64) p.buf_avail[(size_t)e * NA + j] = p.next_avail[(size_t)e * NA + j];
    }
    if (lane < N) { p.rewards[(size_t)e * N + lane] = rew; p.terminals[(size_t)e * N + lane] = term ? 1.f : 0.f; }
    if (lane == 0) {
        p.terminated[e] = term ? 1.f : 0.f; p.truncated[e] = trunc ? 1.f : 0.f; p.done[e] = done ? 1.f : 0.f;
        p.end_step[e] = steps; p.steps[e] = done ? 0 : steps;
        if (p.prev_steps) p.prev_steps[e] = steps - 1;
        if (done && p.totals) {                          // running totals: episodes finished, env steps in them (integer adds)
            atomicAdd(reinterpret_cast<unsigned long long*>(p.totals), 1ull);
            atomicAdd(reinterpret_cast<unsigned long long*>(p.totals) + 1, (unsigned long long)steps);
        }
    }
}

// ------------------------------------------------------------------------------------------------ synthetic frame env
// Atari-shaped input provider (no emulator in this image): uint8 frame stacks drawn from Philox streams keyed by
// (seed, env, step), 16 bytes per draw; reward 1 when the action equals (step count mod A); episodes end with probability
// p_term per step or at max_steps; auto-reset (next_obs = the frame the step returned, cur_obs = what the agent acts on
// next).  cur_obs is a DIFFERENT buffer from the one the agent just acted on (the env alternates two), so the agent can
// store (obs, next_obs) without copying either.  One workgroup per env.
__global__ void __launch_bounds__(256) synth_frames_kernel(xrl_synth_frames_t p, int reset) {
    const int e = blockIdx.x, tid = threadIdx.x, W = p.row_bytes / 16;
    uint4* cur = reinterpret_cast<uint4*>(p.cur_obs + (size_t)e * p.row_bytes);
    if (reset) {
        for (int w = tid; w < W; w += 256) {
            uint32_t r[4];
            philox4x32(p.seed, (uint32_t)e, 0xfffffff0u, (uint32_t)w, r);
            cur[w] = make_uint4(r[0], r[1], r[2], r[3]);
        }
        if (tid == 0) { p.steps[e] = 0; p.done[e] = 0.f; p.end_step[e] = 0; }
        return;
    }
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    const int before = p.steps[e], steps = before + 1;
    uint32_t c[4];
    philox4x32(p.seed, (uint32_t)e, 2u * step, 0x80000000u, c);
    const bool term = u01(c[0]) < p.p_term, trunc = !term && steps >= p.max_steps, done = term || trunc;
    uint4* nxt = reinterpret_cast<uint4*>(p.next_obs + (size_t)e * p.row_bytes);
    for (int w = tid; w < W; w += 256) {
        uint32_t r[4];
        philox4x32(p.seed, (uint32_t)e, 2u * step, (uint32_t)w, r);
        const uint4 v = make_uint4(r[0], r[1], r[2], r[3]);
        nxt[w] = v;
        if (done) {
            philox4x32(p.seed, (uint32_t)e, 2u * step + 1u, (uint32_t)w, r);
            cur[w] = make_uint4(r[0], r[1], r[2], r[3]);
        } else {
            cur[w] = v;
        }
    }
    __syncthreads();                                     // every thread has read steps[e]
    if (tid == 0) {
        p.reward[e] = (p.action[e] == before % p.A) ? 1.f : 0.f;
        p.terminated[e] = term ? 1.f : 0.f; p.truncated[e] = trunc ? 1.f : 0.f; p.done[e] = done ? 1.f : 0.f;
        p.end_step[e] = steps; p.steps[e] = done ? 0 : steps;
    }
}

// ------------------------------------------------------------------------------------------------ post-step bookkeeping

constexpr int POST_THREADS = 1024;

__global__ void __launch_bounds__(POST_THREADS) poststep_kernel(xrl_poststep_t p) {
    __shared__ unsigned long long ended_mask[64];     // up to 4096 envs per pass
    poststep_body<POST_THREADS>(p, ended_mask);
}

// xrl_rollout_poststep of vector step t - 1 followed by xrl_obs_normalize of step t as ONE launch (xrl_post_norm): both are single
// workgroups of 1 024 threads whose work is a few reductions over the env axis, back to back in every vector step of the on-policy loop
// (ppo_agent.py:114-115 after :144-177 of the previous step); neither reads what the other writes.
static_assert(POST_THREADS == RMS_THREADS, "one workgroup runs both bodies");
__global__ void __launch_bounds__(POST_THREADS) post_norm_kernel(xrl_poststep_t post, xrl_rms_t rms) {
    __shared__ unsigned long long ended_mask[64];
    poststep_body<POST_THREADS>(post, ended_mask);
    __syncthreads();
    rms_normalize_body(rms);
}

// epsilon-greedy selection (off_policy.py:138-141): where(rand < eps, randint, greedy)

__global__ void __launch_bounds__(256) egreedy_kernel(xrl_egreedy_t p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    const float* q = p.q + (size_t)e * p.ld;
    int best = 0;
    float bv = q[0];
    for (int j = 1; j < p.A; ++j) if (q[j] > bv) { bv = q[j]; best = j; }     // argmax: first maximal index
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    uint32_t r[4];
    philox4x32(p.seed, (uint32_t)e, step, STREAM_EGREEDY, r);
    const float u = p.uniforms ? p.uniforms[e] : u01(r[0]);
    const int ra = p.randoms ? p.randoms[e] : (int)(r[1] % (uint32_t)p.A);
    const int a = (u < (p.eps_dev ? *p.eps_dev : p.eps)) ? ra : best;
    p.action[e] = a;
    if (p.action_f) p.action_f[e] = (float)a;
}

__global__ void __launch_bounds__(256) marl_select_kernel(xrl_marl_act_t p) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    const int a = marl_select_row(p.q + (size_t)r * p.ld, p.avail ? p.avail + (size_t)r * p.A : nullptr, p.A, p.seed, step, r,
                                  p.eps_dev ? *p.eps_dev : p.eps, p.coin, p.uniforms);
    p.action[r] = a;
    if (p.action_f) p.action_f[r] = (float)a;
}

__global__ void counter_add_kernel(uint32_t* c, uint32_t inc) { *c += inc; }

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_obs_normalize(const xrl_rms_t* params, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_rms_t& p = *params;
    XRL_CHECK_ARG(p.x && p.mean && p.var && p.count && p.n > 0 && p.D > 0 && p.D <= RMS_MAXD);
    hipLaunchKernelGGL(rms_normalize_kernel, dim3(1), dim3(RMS_THREADS), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_policy_sample(const xrl_sample_t* params, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_sample_t& p = *params;
    XRL_CHECK_ARG(p.heads && p.n > 0 && p.A > 0 && p.ld >= p.A);
    XRL_CHECK_ARG(p.ld > p.A || (!p.val_out && !p.bootv_prev));     /* a value column only when someone reads it */
    XRL_CHECK_ARG((p.act_out && p.logp_out) || (!p.act_out && p.bootv_prev));
    XRL_CHECK_ARG(!p.gaussian || p.log_std);
    hipLaunchKernelGGL(policy_sample_kernel, dim3((p.n + 255) / 256), dim3(256), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_cartpole_step(const xrl_cartpole_t* params, int reset, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_cartpole_t& p = *params;
    XRL_CHECK_ARG(p.state && p.steps && p.episodes && p.obs && p.ep_score && p.n > 0);
    if (reset) {
        hipLaunchKernelGGL(cartpole_reset_kernel, dim3((p.n + 255) / 256), dim3(256), 0, as_stream(stream), p);
    } else {
        XRL_CHECK_ARG(p.action && p.next_obs && p.reward && p.terminated && p.truncated && p.stats);
        hipLaunchKernelGGL(cartpole_step_kernel, dim3((p.n + 255) / 256), dim3(256), 0, as_stream(stream), p);
    }
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_synth_control_step(const xrl_synth_ctl_t* params, int reset, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_synth_ctl_t& p = *params;
    XRL_CHECK_ARG(p.state && p.obs && p.steps && p.ep_score && p.n > 0 && p.D > 0 && p.D <= 32 && p.A > 0 && p.A <= 16);
    if (!reset) XRL_CHECK_ARG(p.action && p.next_obs && p.reward && p.terminated && p.truncated && p.stats && p.Amat && p.Bmat);
    hipLaunchKernelGGL(synth_control_kernel, dim3((p.n + 3) / 4), dim3(256), 0, as_stream(stream), p, reset);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_synth_frames_step(const xrl_synth_frames_t* params, int reset, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_synth_frames_t& p = *params;
    XRL_CHECK_ARG(p.cur_obs && p.steps && p.done && p.end_step && p.n > 0 && p.row_bytes > 0 && p.row_bytes % 16 == 0);
    if (!reset) XRL_CHECK_ARG(p.next_obs && p.action && p.reward && p.terminated && p.truncated && p.A > 0 && p.cur_obs != p.next_obs);
    hipLaunchKernelGGL(synth_frames_kernel, dim3(p.n), dim3(256), 0, as_stream(stream), p, reset);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_synth_marl_step(const xrl_synth_marl_t* params, int reset, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_synth_marl_t& p = *params;
    XRL_CHECK_ARG(p.buf_obs && p.buf_state && p.buf_avail && p.steps && p.done && p.end_step && p.n > 0);
    XRL_CHECK_ARG(p.N > 0 && p.N <= 64 && p.O > 0 && p.A > 0 && p.S > 0 && p.N * p.O < 4096 && p.S < 4096 && p.N * p.A < 4096);
    if (!reset) XRL_CHECK_ARG(p.action && p.next_obs && p.next_state && p.next_avail && p.rewards && p.terminals && p.terminated && p.truncated);
    hipLaunchKernelGGL(synth_marl_kernel, dim3((p.n + 3) / 4), dim3(256), 0, as_stream(stream), p, reset);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_rollout_poststep(const xrl_poststep_t* params, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_poststep_t& p = *params;
    // next_obs_norm NULL: the consumer takes the next observations as they are (uint8 frame stacks, agents/ppo_agent.py)
    XRL_CHECK_ARG(p.reward && p.terminated && p.truncated && (p.next_obs || !p.next_obs_norm) && p.rew_out && p.term_out &&
                  p.seg_out && p.ret_track && p.ret_mean && p.ret_var && p.ret_count && p.n > 0 && p.D > 0);
    XRL_CHECK_ARG(!p.use_obsnorm || (p.obs_mean && p.obs_var));
    hipLaunchKernelGGL(poststep_kernel, dim3(1), dim3(POST_THREADS), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_post_norm(const xrl_poststep_t* post, const xrl_rms_t* rms, xrl_stream_t stream) {
    XRL_CHECK_ARG(post != nullptr && rms != nullptr);
    const xrl_poststep_t& p = *post;
    XRL_CHECK_ARG(p.reward && p.terminated && p.truncated && (p.next_obs || !p.next_obs_norm) && p.rew_out && p.term_out &&
                  p.seg_out && p.ret_track && p.ret_mean && p.ret_var && p.ret_count && p.n > 0 && p.D > 0);
    XRL_CHECK_ARG(!p.use_obsnorm || (p.obs_mean && p.obs_var));
    const xrl_rms_t& r = *rms;
    XRL_CHECK_ARG(r.x && r.mean && r.var && r.count && r.n > 0 && r.D > 0 && r.D <= RMS_MAXD);
    hipLaunchKernelGGL(post_norm_kernel, dim3(1), dim3(POST_THREADS), 0, as_stream(stream), p, r);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_egreedy(const xrl_egreedy_t* params, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_egreedy_t& p = *params;
    XRL_CHECK_ARG(p.q && p.action && p.n > 0 && p.A > 0 && p.ld >= p.A);
    hipLaunchKernelGGL(egreedy_kernel, dim3((p.n + 255) / 256), dim3(256), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_marl_select_actions(const xrl_marl_act_t* params, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_marl_act_t& p = *params;
    XRL_CHECK_ARG(p.q && p.action && p.R > 0 && p.A > 0 && p.ld >= p.A);
    hipLaunchKernelGGL(marl_select_kernel, dim3((p.R + 255) / 256), dim3(256), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_counter_add(uint32_t* counter, uint32_t inc, xrl_stream_t stream) {
    XRL_CHECK_ARG(counter != nullptr);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(stream), counter, inc);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

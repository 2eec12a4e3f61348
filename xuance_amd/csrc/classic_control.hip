// Device-resident Pendulum-v1 / MountainCar-v0 / Acrobot-v1 vector envs with the DummyVecEnv auto-reset contract
// (environment/vector_envs/dummy_vec_env.py:65-76), siblings of xrl_cartpole_step (csrc/rollout.hip): the classic-control
// configs of the reference (configs/ppo/classic_control/*.yaml) then run rollout AND update on the device -- their update is
// the shared-trunk family kernel (csrc/ppo_trunk.hip).  One thread per env; physics in classic.h.
#include "common.h"
#include "classic.h"

namespace xrl {

__global__ void __launch_bounds__(256) classic_step_kernel(xrl_classic_t p, int reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n) return;
    const int kind = p.kind, D = classic_obs_dim(kind);
    double* s = p.state + (size_t)e * 4;
    float* o = p.obs + (size_t)e * D;
    if (reset) {
        classic_reset(kind, s, p.seed, e, 0u);
        p.steps[e] = 0; p.episodes[e] = 0; p.ep_score[e] = 0.f;
        classic_observe(kind, s, o);
        return;
    }
    classic_step_one(p, e);
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_classic_step(const xrl_classic_t* params, int reset, xrl_stream_t stream) {
    XRL_CHECK_ARG(params != nullptr);
    const xrl_classic_t& p = *params;
    XRL_CHECK_ARG(p.kind >= CLASSIC_PENDULUM && p.kind <= CLASSIC_ACROBOT && p.n > 0 && p.max_steps > 0);
    XRL_CHECK_ARG(p.state && p.steps && p.episodes && p.obs && p.ep_score);
    if (!reset) {
        XRL_CHECK_ARG(p.next_obs && p.reward && p.terminated && p.truncated && p.stats);
        XRL_CHECK_ARG(p.kind == CLASSIC_PENDULUM ? p.action_f != nullptr : p.action != nullptr);
    }
    hipLaunchKernelGGL(classic_step_kernel, dim3((p.n + 255) / 256), dim3(256), 0, as_stream(stream), p, reset);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

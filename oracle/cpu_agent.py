"""TEST INFRASTRUCTURE ONLY -- CPU port of the PPO rollout+update loop, assembled from oracle/xrl_oracle.py.

Used by (a) ``bench.py``'s ``cpu_baseline`` leg (kind "port": the oracle timed on the GPU box's host cores) and
(b) tests that need a whole-loop CPU run.  It keeps the reference's loop structure -- per-step NumPy buffer writes,
the per-env / per-timestep Python ``finish_path`` loop (memory_tools.py:242-265), per-minibatch fancy-index
``sample`` and one learner update per minibatch (core/on_policy.py:182-205, ppo_agent.py:111-181) -- so its cost
profile is the reference's, with NumPy matmuls standing in for torch CPU ops.  Never imported by xuance_amd/.
"""
import time

import numpy as np

from . import xrl_oracle as o


def orthogonal(rng, shape):
    a = rng.standard_normal(shape)
    q, r = np.linalg.qr(a.T if shape[0] < shape[1] else a)
    q = q * np.sign(np.diag(r))
    return (q.T if shape[0] < shape[1] else q).astype(np.float32)


def init_cartpole_net(rng, obs_dim=4, hidden=128, n_actions=2):
    sd = {}
    for name, (out, inp) in {"representation.model.0": (hidden, obs_dim), "actor.logits.0": (hidden, hidden),
                             "actor.logits.2": (n_actions, hidden), "critic.values.0": (hidden, hidden),
                             "critic.values.2": (1, hidden)}.items():
        sd[name + ".weight"] = orthogonal(rng, (out, inp))
        sd[name + ".bias"] = np.zeros(out, np.float32)
    return sd


class VecCartPole:
    """NumPy CartPole-v1 vector env with DummyVecEnv auto-reset semantics (terminal obs + reset_obs)."""

    def __init__(self, n, rng):
        self.n, self.rng = n, rng
        self.core = o.CartPoleOracle(rng.uniform(-0.05, 0.05, (n, 4)))
        self.buf_obs = self.core.state.astype(np.float32)

    def step(self, actions):
        obs, rew, term, trunc = self.core.step(actions)
        done = term | trunc
        reset_obs = obs.copy()
        if done.any():
            k = int(done.sum())
            self.core.state[done] = self.rng.uniform(-0.05, 0.05, (k, 4))
            self.core.steps[done] = 0
            reset_obs[done] = self.core.state[done].astype(np.float32)
        self.buf_obs = reset_obs
        return obs, rew, term, trunc, reset_obs


def run_ppo_cartpole(n_envs=256, horizon=256, n_rollouts=1, n_epochs=8, n_minibatch=8, seed=1, gamma=0.98, lam=0.95,
                     lr=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, grad_clip_norm=0.5, time_budget_s=None):
    """Returns dict(env_steps, seconds, rollouts, info).  Stops early once time_budget_s is exceeded."""
    rng = np.random.default_rng(seed)
    sd = init_cartpole_net(rng)
    env = VecCartPole(n_envs, rng)
    total_iters = 10 ** 6
    opt = o.AdamOracle(sd, lr=lr, eps=1e-5, total_iters=total_iters)
    cfg = dict(vf_coef=vf_coef, ent_coef=ent_coef, clip_range=clip_range, use_grad_clip=True, grad_clip_norm=grad_clip_norm)
    obs_rms, ret_rms = o.RunningMeanStdOracle((4,)), o.RunningMeanStdOracle(())
    returns = np.zeros(n_envs, np.float32)
    buf = o.OnPolicyBufferOracle((4,), (), n_envs, horizon, gamma=gamma, gae_lam=lam)
    buffer_size = n_envs * horizon
    batch_size = buffer_size // n_minibatch
    obs = env.buf_obs
    info, done_rollouts = {}, 0
    t0 = time.perf_counter()
    while done_rollouts < n_rollouts:
        for _ in range(horizon):
            obs_rms.update(obs)
            obs_n = o.process_observation(obs, obs_rms).astype(np.float32)
            logits, value = o.actor_critic_forward(sd, obs_n)
            acts = o.categorical_sample_icdf(logits, rng.random(n_envs).astype(np.float32))
            logp = o.log_softmax(logits)[np.arange(n_envs), acts]
            next_obs, rew, term, trunc, reset_obs = env.step(acts)
            buf.store(obs_n, acts, o.process_reward(rew, ret_rms), value, term, {"old_logp": logp})
            if buf.full:
                vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms).astype(np.float32))[1]
                for i in range(n_envs):
                    buf.finish_path(0.0 if term[i] else vals[i], i)
                idx = np.arange(buffer_size)
                for _e in range(n_epochs):
                    rng.shuffle(idx)
                    for start in range(0, buffer_size, batch_size):
                        s = buf.sample(idx[start:start + batch_size])
                        info, _ = o.ppo_update(sd, opt, dict(obs=s["obs"], actions=s["actions"], returns=s["returns"],
                                                             advantages=s["advantages"],
                                                             old_logp=s["aux_batch"]["old_logp"]), cfg)
                buf.clear()
            returns = (gamma * returns + rew).astype(np.float32)
            for i in np.flatnonzero(term | trunc):
                ret_rms.update(returns[i:i + 1])
                returns[i] = 0.0
                if term[i]:
                    buf.finish_path(0.0, i)
                else:
                    vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms).astype(np.float32))[1]
                    buf.finish_path(vals[i], i)
            obs = reset_obs
        done_rollouts += 1
        if time_budget_s is not None and time.perf_counter() - t0 > time_budget_s:
            break
    sec = time.perf_counter() - t0
    return dict(env_steps=done_rollouts * buffer_size, seconds=sec, rollouts=done_rollouts,
                info={k: float(v) for k, v in info.items() if not isinstance(v, dict) and (np.isscalar(v) or np.ndim(v) == 0)})

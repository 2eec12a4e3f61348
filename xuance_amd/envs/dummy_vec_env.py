"""Host vector envs with the reference's contracts, for simulators that are plain Python objects (what ``make_envs`` gives
the runner for evaluation, run_drl.py:114,160): sequential stepping, auto-reset, ``infos[i]["reset_obs"]``.

  DummyVecEnv            <-> xuance/environment/vector_envs/dummy/dummy_vec_env.py:7-104
  DummyVecMultiAgentEnv  <-> xuance/environment/vector_envs/dummy/dummy_vec_maenv.py:10-83
  HostSMACLikeEnv        a NumPy stand-in with the SMAC-3m interface (dict observations / rewards / terminations per agent,
                         ``info["state"]``, ``info["avail_actions"]``, ``episode_score`` per agent) for exercising the
                         multi-agent evaluation loop; no simulator is installed in this image.
The training loops of this package keep their environments on the device (envs/cartpole.py, envs/synthetic.py) or behind
ShmSubprocVecEnv; these classes serve ``agent.test(test_envs=...)``."""
import numpy as np

from ..spaces import Box, Discrete, space2shape


class DummyVecEnv:
    def __init__(self, env_fns, env_seed=None):
        self.envs = [fn() if env_seed is None else fn(env_seed=env_seed + i) for i, fn in enumerate(env_fns)]
        env = self.envs[0]
        self.num_envs = len(self.envs)
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.max_episode_steps = getattr(env, "max_episode_steps", None)
        shape = tuple(space2shape(self.observation_space))
        dtype = getattr(self.observation_space, "dtype", np.float32) or np.float32
        self.buf_obs = np.zeros((self.num_envs,) + shape, dtype)
        self.buf_terminated = np.zeros(self.num_envs, bool)
        self.buf_truncated = np.zeros(self.num_envs, bool)
        self.buf_rewards = np.zeros(self.num_envs, np.float32)
        self.buf_info = [{} for _ in range(self.num_envs)]
        self.actions, self.waiting, self.closed = None, False, False

    def reset(self):
        for e, env in enumerate(self.envs):
            self.buf_obs[e], self.buf_info[e] = env.reset()
        return self.buf_obs.copy(), [dict(i) for i in self.buf_info]

    def step_async(self, actions):
        self.actions, self.waiting = actions, True

    def step_wait(self):                                        # dummy_vec_env.py:65-76
        for e, env in enumerate(self.envs):
            obs, self.buf_rewards[e], self.buf_terminated[e], self.buf_truncated[e], info = env.step(self.actions[e])
            self.buf_obs[e], self.buf_info[e] = obs, dict(info)
            if self.buf_terminated[e] or self.buf_truncated[e]:
                self.buf_info[e]["reset_obs"], _ = env.reset()
        self.waiting = False
        return (self.buf_obs.copy(), self.buf_rewards.copy(), self.buf_terminated.copy(), self.buf_truncated.copy(),
                [dict(i) for i in self.buf_info])

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self.closed:
            for env in self.envs:
                env.close()
        self.closed = True


class DummyVecMultiAgentEnv:
    def __init__(self, env_fns, env_seed=None):
        self.envs = [fn() if env_seed is None else fn(env_seed=env_seed + i) for i, fn in enumerate(env_fns)]
        env = self.envs[0]
        self.num_envs = len(self.envs)
        self.agents = self.agent_keys = list(env.agents)
        self.num_agents = len(self.agents)
        self.state_space, self.observation_space, self.action_space = env.state_space, env.observation_space, env.action_space
        self.max_episode_steps = env.max_episode_steps
        self.buf_state = [np.zeros(space2shape(self.state_space)) for _ in range(self.num_envs)]
        self.buf_obs = [{} for _ in range(self.num_envs)]
        self.buf_avail_actions = [{} for _ in range(self.num_envs)]
        self.buf_info = [{} for _ in range(self.num_envs)]
        self.actions, self.waiting, self.closed = None, False, False

    def reset(self):                                            # dummy_vec_maenv.py:33-42
        for e, env in enumerate(self.envs):
            self.buf_obs[e], self.buf_info[e] = env.reset()
            self.buf_state[e] = self.buf_info[e]["state"]
            self.buf_avail_actions[e] = self.buf_info[e]["avail_actions"]
        return list(self.buf_obs), list(self.buf_info)

    def step_async(self, actions):
        self.actions, self.waiting = actions, True

    def step_wait(self):                                        # dummy_vec_maenv.py:62-83
        rew, term, trunc = [{} for _ in self.envs], [{} for _ in self.envs], [False for _ in self.envs]
        for e, env in enumerate(self.envs):
            self.buf_obs[e], rew[e], term[e], trunc[e], self.buf_info[e] = env.step(self.actions[e])
            self.buf_avail_actions[e] = self.buf_info[e]["avail_actions"]
            self.buf_state[e] = self.buf_info[e]["state"]
            if all(term[e].values()) or trunc[e]:
                obs_reset, info_reset = env.reset()
                self.buf_info[e]["reset_obs"] = obs_reset
                self.buf_info[e]["reset_avail_actions"] = info_reset["avail_actions"]
                self.buf_info[e]["reset_state"] = info_reset["state"]
        self.waiting = False
        return list(self.buf_obs), rew, term, trunc, list(self.buf_info)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self.closed:
            for env in self.envs:
                env.close()
        self.closed = True


class HostSMACLikeEnv:
    """SMAC-3m-shaped host env (3 agents, obs 30, state 48, 9 actions with an availability mask, 60-step episodes): the
    reward favours ONE action per step (the index the state's first component points to) when it is available, so a policy
    is scoreable; dynamics are a seeded random walk."""
    n_agents, obs_dim, state_dim, n_actions, max_episode_steps = 3, 30, 48, 9, 60
    strict_actions = True            # an unavailable action is an error (False: it merely earns nothing)

    def __init__(self, env_seed=None):
        self.agents = [f"agent_{i}" for i in range(self.n_agents)]
        self.observation_space = {k: Box(-np.inf, np.inf, (self.obs_dim,), np.float32) for k in self.agents}
        self.action_space = {k: Discrete(self.n_actions) for k in self.agents}
        self.state_space = Box(-np.inf, np.inf, (self.state_dim,), np.float32)
        self.rng = np.random.default_rng(env_seed)
        self.steps, self.score = 0, None

    def _emit(self):
        self.state = self.rng.standard_normal(self.state_dim).astype(np.float32)
        obs = {k: self.rng.standard_normal(self.obs_dim).astype(np.float32) for k in self.agents}
        avail = {}
        for k in self.agents:
            a = self.rng.random(self.n_actions) < 0.7
            a[0] = True
            avail[k] = a.astype(np.float32)
        self.avail = avail
        return obs

    def reset(self):
        self.steps, self.score = 0, {k: 0.0 for k in self.agents}
        obs = self._emit()
        return obs, {"state": self.state, "avail_actions": self.avail, "agent_mask": {k: True for k in self.agents},
                     "episode_step": 0, "episode_score": dict(self.score)}

    def step(self, actions):
        target = int(abs(self.state[0]) * 3) % self.n_actions
        rew = {}
        for k in self.agents:
            ok = self.avail[k][int(actions[k])] > 0
            assert ok or not self.strict_actions, "unavailable action chosen"
            rew[k] = 1.0 if (ok and int(actions[k]) == target) else 0.0
            self.score[k] += rew[k]
        self.steps += 1
        term = bool(self.rng.random() < 0.02)
        trunc = (not term) and self.steps >= self.max_episode_steps
        obs = self._emit()
        info = {"state": self.state, "avail_actions": self.avail, "agent_mask": {k: True for k in self.agents},   # wrapper.py:160-178
                "episode_step": self.steps, "episode_score": dict(self.score)}
        return obs, rew, {k: term for k in self.agents}, trunc, info

    def close(self):
        pass

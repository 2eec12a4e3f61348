"""Device-resident CartPole-v1 vector environment (xrl_cartpole_step).

Presents the VecEnv surface the reference agents use (``num_envs``, ``observation_space``, ``action_space``,
``buf_obs``, ``reset()``, ``step(actions)``, ``close()``; xuance/environment/vector_envs/vector_env.py:17-78,
dummy_vec_env.py:7-104) plus ``step_device()``, which enqueues one kernel and leaves every output in HBM so the
on-policy rollout never leaves the GPU.  Auto-reset follows DummyVecEnv.step_wait (dummy_vec_env.py:65-76): the
returned observation of a finished episode is the terminal one and ``infos[i]["reset_obs"]`` / ``buf_obs`` hold
the first observation of the next episode.
"""
import numpy as np
import torch

from .. import ops
from ..spaces import Box, Discrete


class DeviceCartPoleVecEnv:
    max_episode_steps = 500

    def __init__(self, num_envs, seed=1, device="cuda"):
        self.num_envs, self.seed, self.device = int(num_envs), int(seed), device
        high = np.array([4.8, np.finfo(np.float32).max, 0.41887903, np.finfo(np.float32).max], np.float32)
        self.observation_space = Box(-high, high, (4,), np.float32)
        self.action_space = Discrete(2)
        n, dev = self.num_envs, device
        self.state = torch.zeros(n, 4, dtype=torch.float64, device=dev)
        self.steps = torch.zeros(n, dtype=torch.int32, device=dev)
        self.episodes = torch.zeros(n, dtype=torch.int32, device=dev)
        self.action = torch.zeros(n, dtype=torch.int32, device=dev)
        self.buf_obs = torch.zeros(n, 4, device=dev)          # observation the agent acts on next
        self.next_obs = torch.zeros(n, 4, device=dev)         # observation returned by the last step (pre-reset)
        self.reward = torch.zeros(n, device=dev)
        self.terminated = torch.zeros(n, device=dev)
        self.truncated = torch.zeros(n, device=dev)
        self.ep_score = torch.zeros(n, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self.max_episode_steps = DeviceCartPoleVecEnv.max_episode_steps

    def _kw(self):
        return dict(state=self.state, steps=self.steps, episodes=self.episodes, action=self.action, obs=self.buf_obs,
                    next_obs=self.next_obs, reward=self.reward, terminated=self.terminated, truncated=self.truncated,
                    ep_score=self.ep_score, stats=self.stats, n=self.num_envs, max_steps=self.max_episode_steps,
                    seed=self.seed)

    def reset(self):
        ops.cartpole_step(reset=True, **self._kw())
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self):
        """One vector step from ``self.action`` (int32 device tensor); nothing is copied to the host."""
        ops.cartpole_step(**self._kw())

    def step(self, actions):
        """Host-compatible step (synchronises): returns NumPy arrays like DummyVecEnv.step_wait."""
        self.action.copy_(torch.as_tensor(np.asarray(actions)).to(torch.int32))
        self.step_device()
        obs = self.next_obs.cpu().numpy()
        term = self.terminated.cpu().numpy() > 0
        trunc = self.truncated.cpu().numpy() > 0
        reset_obs = self.buf_obs.cpu().numpy()
        infos = [{"reset_obs": reset_obs[i]} if (term[i] or trunc[i]) else {} for i in range(self.num_envs)]
        return obs, self.reward.cpu().numpy(), term, trunc, infos

    def episode_stats(self):
        """(finished episodes, mean score, mean length) since construction."""
        s = self.stats.cpu().numpy()
        n = max(s[0], 1.0)
        return int(s[0]), float(s[1] / n), float(s[2] / n)

    def close(self):
        pass

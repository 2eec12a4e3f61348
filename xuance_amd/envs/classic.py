"""Device-resident Pendulum-v1, MountainCar-v0 and Acrobot-v1 vector envs (xrl_classic_step, csrc/classic_control.hip): the other
environments of the reference's configs/ppo/classic_control/*.yaml next to CartPole-v1 (envs/cartpole.py), same surface as
DummyVecEnv (environment/vector_envs/dummy_vec_env.py:8-110: buf_obs, auto-reset with the terminal observation returned and the reset
observation kept) plus step_device() for the agents' device loops.  The dynamics are Gymnasium's published ones (third-party, not in
the image): csrc/classic.h restates them, oracle/xrl_oracle.py: PendulumOracle / MountainCarOracle / AcrobotOracle are the NumPy twins
the tests compare with.  Time limits as registered by Gymnasium: 200 / 200 / 500 steps."""
import numpy as np
import torch

from .. import ops
from ..spaces import Box, Discrete


class _DeviceClassicVecEnv:
    kind, obs_dim, max_episode_steps, graph_safe = 0, 0, 0, True

    def __init__(self, num_envs, seed=1, device="cuda", max_episode_steps=None):
        self.num_envs, self.seed, self.device = int(num_envs), int(seed), device
        if max_episode_steps is not None:
            self.max_episode_steps = int(max_episode_steps)
        n, dev, D = self.num_envs, device, self.obs_dim
        self.state = torch.zeros(n, 4, dtype=torch.float64, device=dev)
        self.steps = torch.zeros(n, dtype=torch.int32, device=dev)
        self.episodes = torch.zeros(n, dtype=torch.int32, device=dev)
        self.buf_obs = torch.zeros(n, D, device=dev)          # observation the agent acts on next
        self.next_obs = torch.zeros(n, D, device=dev)         # observation returned by the last step (pre-reset)
        self.reward = torch.zeros(n, device=dev)
        self.terminated = torch.zeros(n, device=dev)
        self.truncated = torch.zeros(n, device=dev)
        self.ep_score = torch.zeros(n, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self.action = self._action_tensor()

    def _action_tensor(self):
        return torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)

    def _kw(self):
        cont = self.action.dtype == torch.float32
        return dict(state=self.state, steps=self.steps, episodes=self.episodes, action=None if cont else self.action,
                    action_f=self.action if cont else None, obs=self.buf_obs, next_obs=self.next_obs, reward=self.reward,
                    terminated=self.terminated, truncated=self.truncated, ep_score=self.ep_score, stats=self.stats, n=self.num_envs,
                    kind=self.kind, max_steps=self.max_episode_steps, seed=self.seed)

    def reset(self):
        ops.classic_step(reset=True, **self._kw())
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self):
        """One vector step from ``self.action`` (a device tensor); nothing is copied to the host."""
        ops.classic_step(**self._kw())

    def step(self, actions):
        """Host-compatible step (synchronises): returns NumPy arrays like DummyVecEnv.step_wait."""
        self.action.copy_(torch.as_tensor(np.asarray(actions)).reshape(self.action.shape).to(self.action.dtype))
        self.step_device()
        obs = self.next_obs.cpu().numpy()
        term, trunc = self.terminated.cpu().numpy() > 0, self.truncated.cpu().numpy() > 0
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


class DevicePendulumVecEnv(_DeviceClassicVecEnv):
    """Pendulum-v1: obs (cos theta, sin theta, theta_dot), Box(-2, 2, (1,)) torque, 200 steps, never terminates."""
    kind, obs_dim, max_episode_steps = 1, 3, 200

    def __init__(self, num_envs, seed=1, device="cuda", max_episode_steps=None):
        super().__init__(num_envs, seed, device, max_episode_steps)
        high = np.array([1.0, 1.0, 8.0], np.float32)
        self.observation_space, self.action_space = Box(-high, high, (3,), np.float32), Box(-2.0, 2.0, (1,), np.float32)

    def _action_tensor(self):
        return torch.zeros(self.num_envs, 1, dtype=torch.float32, device=self.device)


class DeviceMountainCarVecEnv(_DeviceClassicVecEnv):
    """MountainCar-v0: obs (position, velocity), Discrete(3), reward -1 per step, 200 steps."""
    kind, obs_dim, max_episode_steps = 2, 2, 200

    def __init__(self, num_envs, seed=1, device="cuda", max_episode_steps=None):
        super().__init__(num_envs, seed, device, max_episode_steps)
        self.observation_space = Box(np.array([-1.2, -0.07], np.float32), np.array([0.6, 0.07], np.float32), (2,), np.float32)
        self.action_space = Discrete(3)


class DeviceAcrobotVecEnv(_DeviceClassicVecEnv):
    """Acrobot-v1: obs (cos t1, sin t1, cos t2, sin t2, dt1, dt2), Discrete(3) torque -1 / 0 / +1, RK4 over 0.2 s, 500 steps."""
    kind, obs_dim, max_episode_steps = 3, 6, 500

    def __init__(self, num_envs, seed=1, device="cuda", max_episode_steps=None):
        super().__init__(num_envs, seed, device, max_episode_steps)
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * np.pi, 9 * np.pi], np.float32)
        self.observation_space, self.action_space = Box(-high, high, (6,), np.float32), Discrete(3)

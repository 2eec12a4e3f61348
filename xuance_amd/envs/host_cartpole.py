"""CartPole-v1 as a plain host (NumPy) environment with the gymnasium step/reset contract: the kind of Python simulator
the reference runs behind DummyVecEnv / SubprocVecEnv.  Physics: Barto, Sutton & Anderson 1983 with the constants and
explicit-Euler update published with Gymnasium's classic_control/cartpole.py (not part of the reference tree) -- the
same equations as csrc/cartpole.h.  Used to exercise ShmSubprocVecEnv with real worker processes."""
import math

import numpy as np

from ..spaces import Box, Discrete


class NumpyCartPoleEnv:
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_threshold, x_threshold = 12 * 2 * math.pi / 360, 2.4
    max_episode_steps = 500

    def __init__(self, env_seed=None):
        high = np.array([4.8, np.finfo(np.float32).max, 0.41887903, np.finfo(np.float32).max], np.float32)
        self.observation_space, self.action_space = Box(-high, high, (4,), np.float32), Discrete(2)
        self.rng = np.random.default_rng(env_seed)
        self.state, self.steps, self.score = None, 0, 0.0

    def reset(self, seed=None):
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.state = self.rng.uniform(-0.05, 0.05, 4)
        self.steps, self.score = 0, 0.0
        return self.state.astype(np.float32), {}

    def step(self, action):
        x, xd, th, thd = self.state
        force = self.force_mag if int(action) == 1 else -self.force_mag
        ct, st = math.cos(th), math.sin(th)
        total_mass, pml = self.masspole + self.masscart, self.masspole * self.length
        temp = (force + pml * thd * thd * st) / total_mass
        thacc = (self.gravity * st - ct * temp) / (self.length * (4.0 / 3.0 - self.masspole * ct * ct / total_mass))
        xacc = temp - pml * thacc * ct / total_mass
        x, xd, th, thd = x + self.tau * xd, xd + self.tau * xacc, th + self.tau * thd, thd + self.tau * thacc
        self.state = np.array([x, xd, th, thd])
        self.steps += 1
        self.score += 1.0
        terminated = bool(x < -self.x_threshold or x > self.x_threshold or th < -self.theta_threshold or th > self.theta_threshold)
        truncated = self.steps >= self.max_episode_steps
        return self.state.astype(np.float32), 1.0, terminated, truncated, {"episode_step": self.steps, "episode_score": self.score}

    def close(self):
        pass

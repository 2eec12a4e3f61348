"""Recorded trajectories as device-resident providers: a vector env whose ``step_device()`` hands out step k of a tape that lives in
HBM instead of running a simulator.  Same surface as the other device providers of this package (envs/cartpole.py,
envs/synthetic.py: ``buf_obs``, ``next_obs``, ``reward``, ``terminated``, ``truncated``, ``action``, ``step_device(offset)``,
``advance(k)``), same auto-reset contract as the reference's DummyVecEnv.step_wait (dummy_vec_env.py:65-76: ``next_obs`` is the
terminal observation of a finished episode, ``buf_obs`` the first observation of the next one).  The actions the agent writes are
ignored -- a tape only makes sense with the actions it was recorded with, which the agents take through their supplied-randomness
hooks (``PPO_Agent.action_noise``, ``DQN_Agent.explore_tape``, ``QMIX_Agents.explore_tape``).

Uses: replaying a run of ANOTHER implementation of the loop on the same simulator outputs (tests/test_gpu_agent_replay.py replays
runs of the reference's own agents recorded by oracle/make_golden_agents.py) and feeding logged data to the device loops.

The step index is a device counter (+ a static offset inside a captured rollout; the counter ticks once per rollout through
``advance``), the rows are fetched by index_select on that counter, so a whole rollout over a tape can be captured into a hipGraph
and replayed on the following stretch of the tape."""
import numpy as np
import torch

from ..spaces import Box, Discrete


def _dev(x, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype).to(device).contiguous()


class RecordedVecEnv:
    graph_safe = True

    def __init__(self, obs0, next_obs, rewards, terminated, truncated, reset_obs, observation_space=None, action_space=None,
                 max_episode_steps=None, device="cuda"):
        """obs0 [n, *obs]: what the loop acts on first; next_obs [S, n, *obs], rewards / terminated / truncated [S, n]: what step k
        returned; reset_obs [S, n, *obs]: ``infos[i]["reset_obs"]`` of the envs that finished in step k (other rows ignored)."""
        next_obs = np.asarray(next_obs)
        S, n = next_obs.shape[:2]
        self.num_envs, self.n_steps, self.device = int(n), int(S), device
        self.obs_shape = tuple(next_obs.shape[2:])
        odt = torch.uint8 if next_obs.dtype == np.uint8 else torch.float32
        self.observation_space = observation_space or Box(-np.inf, np.inf, self.obs_shape, np.uint8 if odt == torch.uint8 else np.float32)
        self.action_space = action_space or Discrete(2)
        self.max_episode_steps = max_episode_steps
        done = (np.asarray(terminated) > 0) | (np.asarray(truncated) > 0)
        cur = np.where(done.reshape((S, n) + (1,) * len(self.obs_shape)), np.asarray(reset_obs), next_obs)   # dummy_vec_env.py:71-74 + the
        self._cur = _dev(np.concatenate([np.asarray(obs0)[None], cur]), odt, device)                           # agent's obs[i] = reset_obs
        self._next = _dev(next_obs, odt, device)
        self._rew = _dev(rewards, torch.float32, device)
        self._term = _dev(np.asarray(terminated) > 0, torch.float32, device)
        self._trunc = _dev(np.asarray(truncated) > 0, torch.float32, device)
        self.buf_obs = torch.zeros((n,) + self.obs_shape, dtype=odt, device=device)
        self.next_obs = torch.zeros_like(self.buf_obs)
        self.reward = torch.zeros(n, device=device)
        self.terminated = torch.zeros(n, device=device)
        self.truncated = torch.zeros(n, device=device)
        A = getattr(self.action_space, "n", None)
        self.action = torch.zeros(n, dtype=torch.int32, device=device) if A is not None else \
            torch.zeros((n,) + tuple(self.action_space.shape), device=device)
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=device)
        self._host_step = 0
        self._idx = {}                                                 # static offset -> its index tensor (allocated outside captures)
        self._idx_host = torch.zeros(1, dtype=torch.int64, device=device)

    def prepare(self, horizon):
        """Allocate the per-offset index tensors of a captured rollout of `horizon` steps (nothing may allocate inside a capture)."""
        for t in range(int(horizon)):
            self._idx.setdefault(t, torch.zeros(1, dtype=torch.int64, device=self.device))

    def reset(self):
        self.buf_obs.copy_(self._cur[0])
        self.step_counter.zero_()
        self._host_step = 0
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def _emit(self, idx):
        n = self.num_envs
        torch.index_select(self._next, 0, idx, out=self.next_obs.view((1, n) + self.obs_shape))
        torch.index_select(self._rew, 0, idx, out=self.reward.view(1, n))
        torch.index_select(self._term, 0, idx, out=self.terminated.view(1, n))
        torch.index_select(self._trunc, 0, idx, out=self.truncated.view(1, n))
        torch.add(idx, 1, out=idx)
        torch.index_select(self._cur, 0, idx, out=self.buf_obs.view((1, n) + self.obs_shape))

    def step_device(self, offset=None):
        if offset is None:                                             # eager loops: the host knows the step index
            assert self._host_step < self.n_steps, "the tape is exhausted"
            self._idx_host.fill_(self._host_step)
            self._host_step += 1
            return self._emit(self._idx_host)
        idx = self._idx.get(int(offset))
        if idx is None:
            idx = self._idx[int(offset)] = torch.zeros(1, dtype=torch.int64, device=self.device)
        torch.add(self.step_counter, int(offset), out=idx)
        self._emit(idx)

    def advance(self, k):
        self.step_counter.add_(int(k))
        self._host_step += int(k)

    def step(self, actions=None):
        """Host-compatible step (synchronises), DummyVecEnv.step_wait's return values."""
        self.step_device()
        term, trunc = self.terminated.cpu().numpy() > 0, self.truncated.cpu().numpy() > 0
        reset = self.buf_obs.cpu().numpy()
        infos = [{"reset_obs": reset[i]} if (term[i] or trunc[i]) else {} for i in range(self.num_envs)]
        return self.next_obs.cpu().numpy(), self.reward.cpu().numpy(), term, trunc, infos

    def close(self):
        pass

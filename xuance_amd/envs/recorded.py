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
                 max_episode_steps=None, device="cuda", restart=None):
        """obs0 [n, *obs]: what the loop acts on first; next_obs [S, n, *obs], rewards / terminated / truncated [S, n]: what step k
        returned; reset_obs [S, n, *obs]: ``infos[i]["reset_obs"]`` of the envs that finished in step k (other rows ignored).
        restart [S, n] (default terminated | truncated): the envs whose next acted-on observation is reset_obs -- the Atari mode of the
        reference's loops restarts on truncation only (off_policy.py:240-242, ppo_agent.py:150-151)."""
        next_obs = np.asarray(next_obs)
        S, n = next_obs.shape[:2]
        self.num_envs, self.n_steps, self.device = int(n), int(S), device
        self.obs_shape = tuple(next_obs.shape[2:])
        odt = torch.uint8 if next_obs.dtype == np.uint8 else torch.float32
        self.observation_space = observation_space or Box(-np.inf, np.inf, self.obs_shape, np.uint8 if odt == torch.uint8 else np.float32)
        self.action_space = action_space or Discrete(2)
        self.max_episode_steps = max_episode_steps
        done = ((np.asarray(terminated) > 0) | (np.asarray(truncated) > 0)) if restart is None else (np.asarray(restart) > 0)
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


class TapeCartPoleVecEnv:
    """A recorded run of a CartPole-class vector env (4-d observations, two actions, reward 1 per step) as the provider of the
    ONE-LAUNCH rollout kernel (csrc/rollout_actor.hip, xrl_rollout_run_t.tape_*): the resident workgroups read what `envs.step()`
    returned -- next observation, terminated / truncated, `infos[i]["reset_obs"]` -- from this tape instead of integrating the
    in-kernel physics; the agent supplies the recorded action draws through `PPO_Agent.set_action_noise`.  Everything else of the
    timed rollout path runs unchanged, which pins it to a run of the reference's own PPO_Agent.train (tests/test_gpu_agent_replay.py;
    reference: ppo_agent.py:111-181, on_policy.py:128-169, dummy_vec_env.py:65-76).  Duck-types envs/cartpole.py's provider (the
    simulator-state tensors exist, unused by the tape instances of the kernel); `tape_pos` is the device counter of the tape row of the
    next rollout's first vector step (PPO_Agent advances it by horizon_size per rollout, inside the captured rollout graph)."""
    graph_safe = True
    is_cartpole_tape = True

    def __init__(self, obs0, next_obs, rewards, terminated, truncated, reset_obs, max_episode_steps=500, device="cuda"):
        next_obs = np.asarray(next_obs, np.float32)
        S, n, D = next_obs.shape
        assert D == 4, "the CartPole class: 4-d observations"
        assert np.all(np.asarray(rewards) == 1.0), "the one-launch CartPole rollout has the class's reward (1 per step) built in"
        self.num_envs, self.n_steps, self.device, self.seed = int(n), int(S), device, 0
        self.max_episode_steps = int(max_episode_steps)
        self.observation_space = Box(-np.inf, np.inf, (4,), np.float32)
        self.action_space = Discrete(2)
        f32 = lambda x: _dev(x, torch.float32, device)
        self._obs0 = f32(obs0)
        self.tape = dict(next_obs=f32(next_obs), reset_obs=f32(np.asarray(reset_obs, np.float32)),
                         term=f32(np.asarray(terminated) > 0), trunc=f32(np.asarray(truncated) > 0))
        self.tape_pos = torch.zeros(1, dtype=torch.int32, device=device)
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)
        self.state, self.steps, self.episodes = z(n, 4, dt=torch.float64), z(n, dt=torch.int32), z(n, dt=torch.int32)
        self.action, self.buf_obs, self.next_obs = z(n, dt=torch.int32), z(n, 4), z(n, 4)
        self.reward, self.terminated, self.truncated, self.ep_score = z(n), z(n), z(n), z(n)
        self.stats = z(4, dt=torch.float64)

    def reset(self):
        self.buf_obs.copy_(self._obs0)
        self.tape_pos.zero_()
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self, offset=None):
        raise NotImplementedError("TapeCartPoleVecEnv feeds the one-launch rollout kernel only (use RecordedVecEnv for the launches per vector step)")

    def close(self):
        pass


class TapeControlVecEnv:
    """TapeCartPoleVecEnv's twin for the two-branch Gaussian class D-256-256-{A | 1} (configs/ppo/mujoco.yaml; csrc/rollout_wide.hip,
    xrl_rollout_wide_t.tape_*): a recorded run of a continuous-control vector env (observations [D <= 20], Box(A <= 8) actions, any
    rewards, terminations and truncations) as the provider of the one-launch rollout kernel xrl_rollout_wide_run; the recorded action
    draws go in through `PPO_Agent.set_action_noise` as standard normals.  Duck-types envs/synthetic.py: SyntheticMujocoVecEnv."""
    graph_safe = True
    is_control_tape = True

    def __init__(self, obs0, next_obs, rewards, terminated, truncated, reset_obs, act_dim, max_episode_steps=1000, device="cuda"):
        from .. import ops
        self._ops = ops
        next_obs = np.asarray(next_obs, np.float32)
        S, n, D = next_obs.shape
        self.num_envs, self.n_steps, self.device, self.seed = int(n), int(S), device, 0
        self.max_episode_steps = int(max_episode_steps)
        self.observation_space = Box(-np.inf, np.inf, (D,), np.float32)
        self.action_space = Box(-1.0, 1.0, (int(act_dim),), np.float32)
        f32 = lambda x: _dev(x, torch.float32, device)
        self._obs0 = f32(obs0)
        self.tape = dict(next_obs=f32(next_obs), reset_obs=f32(np.asarray(reset_obs, np.float32)), rew=f32(rewards),
                         term=f32(np.asarray(terminated) > 0), trunc=f32(np.asarray(truncated) > 0))
        self.tape_pos = torch.zeros(1, dtype=torch.int32, device=device)
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)
        self.A, self.B, self.state = z(D, D), z(int(act_dim), D), z(n, D)
        self.steps, self.ep_score, self.stats = z(n, dt=torch.int32), z(n), z(4, dt=torch.float64)
        self.step_counter = z(1, dt=torch.int32)
        self.buf_obs, self.next_obs = z(n, D), z(n, D)
        self.action = z(n, int(act_dim))
        self.reward, self.terminated, self.truncated = z(n), z(n), z(n)

    def reset(self):
        self.buf_obs.copy_(self._obs0)
        self.tape_pos.zero_()
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def advance(self, k):
        self._ops.counter_add(self.step_counter, int(k))
        self._ops.counter_add(self.tape_pos, int(k))            # the next rollout reads the following stretch of the tape

    def step_device(self, offset=None):
        raise NotImplementedError("TapeControlVecEnv feeds the one-launch rollout kernel only (use RecordedVecEnv for the launches per vector step)")

    def close(self):
        pass


class RecordedMultiAgentVecEnv:
    """The multi-agent twin: a recorded run of a vector env with the reference's multi-agent contract (dummy_vec_maenv.py:33-83 --
    per-agent observations, global state, availability masks, per-agent rewards / terminated flags, one truncated flag per env,
    `reset_obs` / `reset_state` / `reset_avail_actions` of finished envs, `episode_step`) played back with the surface of
    envs/synthetic.py: SyntheticSMACVecEnv (buf_obs [n, N, O], buf_state [n, S], buf_avail [n, N, A], next_*, rewards / terminals
    [n, N], agent_mask, done, end_step, steps).  ``reset()`` loads the next recorded reset (the episode loop of recurrent agents
    resets its envs at the start of every run_episodes call, off_policy_marl.py:436); eager loops only."""
    graph_safe = False

    def __init__(self, resets, next_obs, next_state, next_avail, rewards, terminals, truncations, agent_mask, reset_obs, reset_state,
                 reset_avail, episode_step=None, agent_keys=None, max_episode_steps=None, device="cuda"):
        """resets: [dict(obs [n, N, O], state [n, S], avail [n, N, A], at=tape position)] in call order; everything else [S, n, ...]
        per recorded vector step (reset_* rows matter for finished envs only; episode_step = infos[i]["episode_step"])."""
        next_obs = np.asarray(next_obs, np.float32)
        S, n, N, O = next_obs.shape
        A, Sd = np.asarray(next_avail).shape[-1], np.asarray(next_state).shape[-1]
        self.num_envs, self.n_agents, self.obs_dim, self.state_dim, self.n_actions = n, N, O, Sd, A
        self.n_steps, self.device, self.max_episode_steps = S, device, max_episode_steps
        self.agent_keys = self.agents = list(agent_keys or [f"agent_{i}" for i in range(N)])
        self.observation_space = {k: Box(-np.inf, np.inf, (O,), np.float32) for k in self.agent_keys}
        self.action_space = {k: Discrete(A) for k in self.agent_keys}
        self.state_space = Box(-np.inf, np.inf, (Sd,), np.float32)
        term = np.asarray(terminals) > 0
        done = term.all(-1) | (np.asarray(truncations) > 0)
        d3, d2 = done[:, :, None, None], done[:, :, None]
        f32 = lambda x: _dev(x, torch.float32, device)
        self._next = (f32(next_obs), f32(next_state), f32(next_avail))
        self._cur = (f32(np.where(d3, reset_obs, next_obs)), f32(np.where(d2, reset_state, next_state)), f32(np.where(d3, reset_avail, next_avail)))
        self._rew, self._term, self._mask, self._done = f32(rewards), f32(term), f32(agent_mask), f32(done)
        es = np.asarray(episode_step, np.int64) if episode_step is not None else np.zeros((S, n), np.int64)
        self._end = _dev(es, torch.int32, device)
        self._steps_next = _dev(np.where(done, 0, es), torch.int32, device)
        self._resets = [dict(obs=f32(r["obs"]), state=f32(r["state"]), avail=f32(r["avail"]), at=int(r["at"])) for r in resets]
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)
        self.buf_obs, self.buf_state, self.buf_avail = z(n, N, O), z(n, Sd), z(n, N, A)
        self.next_obs, self.next_state, self.next_avail = z(n, N, O), z(n, Sd), z(n, N, A)
        self.rewards, self.terminals, self.agent_mask = z(n, N), z(n, N), torch.ones(n, N, device=device)
        self.done, self.end_step, self.steps = z(n), z(n, dt=torch.int32), z(n, dt=torch.int32)
        self.action = z(n, N, dt=torch.int32)
        self._pos, self._n_resets = 0, 0

    def reset(self):
        r = self._resets[self._n_resets]
        assert r["at"] == self._pos, f"reset #{self._n_resets} was recorded at tape position {r['at']}, the loop is at {self._pos}"
        self._n_resets += 1
        self.buf_obs.copy_(r["obs"]); self.buf_state.copy_(r["state"]); self.buf_avail.copy_(r["avail"])
        self.steps.zero_(); self.done.zero_()
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self):
        k = self._pos
        assert k < self.n_steps, "the tape is exhausted"
        self._pos += 1
        for dst, src in zip((self.next_obs, self.next_state, self.next_avail), self._next):
            dst.copy_(src[k])
        for dst, src in zip((self.buf_obs, self.buf_state, self.buf_avail), self._cur):
            dst.copy_(src[k])
        self.rewards.copy_(self._rew[k]); self.terminals.copy_(self._term[k]); self.agent_mask.copy_(self._mask[k])
        self.done.copy_(self._done[k]); self.end_step.copy_(self._end[k]); self.steps.copy_(self._steps_next[k])

    def close(self):
        pass

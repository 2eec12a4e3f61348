"""Shape-faithful synthetic vector environments that live on the device (no simulator is installed in this image:
SURVEY.md section 8d).  They emit exactly the tensors the agents of BASELINE configs C3-C5 consume -- Atari-shaped uint8
frame stacks, MuJoCo-shaped float vectors with continuous actions, SMAC-3m-shaped multi-agent observations / global
state / action masks -- with cheap deterministic dynamics so that rollouts, buffers and learners are exercised at the
real shapes.  They are input providers, not part of the measured hot path."""
import numpy as np
import torch

from ..spaces import Box, Discrete


class _Base:
    graph_safe = False      # torch RNG ops inside step_device(): do not capture these envs into a hipGraph

    def __init__(self, num_envs, seed, device, max_episode_steps):
        self.num_envs, self.device, self.max_episode_steps = int(num_envs), device, int(max_episode_steps)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.steps = torch.zeros(self.num_envs, dtype=torch.int32, device=device)
        self.terminated = torch.zeros(self.num_envs, device=device)
        self.truncated = torch.zeros(self.num_envs, device=device)
        self.reward = torch.zeros(self.num_envs, device=device)

    def _end_of_step(self, p_term):
        self.steps += 1
        self.terminated = (torch.rand(self.num_envs, device=self.device, generator=self.gen) < p_term).float()
        self.truncated = ((self.steps >= self.max_episode_steps) & (self.terminated == 0)).float()
        done = (self.terminated + self.truncated) > 0
        self.done = done.float()                    # episode ended in this step ...
        self.end_step = self.steps.clone()          # ... after this many steps (info["episode_step"])
        self.steps[done] = 0
        return done

    def close(self):
        pass


class SyntheticAtariVecEnv(_Base):
    """84x84x4 uint8 frame stacks, Discrete(4) (Breakout-shaped, configs/dqn/atari.yaml:7-8).  One native launch per
    vector step (xrl_synth_frames_step).  `double_buffered`: step_device() leaves the tensor that was `buf_obs` before the
    call untouched and rebinds `buf_obs` to the other of two buffers, so an agent can hand (obs, next_obs) to the replay
    ring without copying the 28 KB frames."""
    graph_safe = False          # buf_obs alternates between two tensors: addresses change from step to step ...
    graph_safe_even = True      # ... with period 2: a captured sequence of an EVEN number of steps (step_device(offset=t) + advance)
    double_buffered = True      #     replays on the same addresses

    def __init__(self, num_envs, seed=1, device="cuda", n_actions=4, max_episode_steps=1000, p_term=0.002):
        super().__init__(num_envs, seed, device, max_episode_steps)
        self.seed, self.p_term = int(seed), float(p_term)
        self.observation_space = Box(0, 255, (84, 84, 4), np.uint8)
        self.action_space = Discrete(n_actions)
        self._bufs = [torch.zeros(self.num_envs, 84, 84, 4, dtype=torch.uint8, device=device) for _ in range(2)]
        self._cur = 0
        self.buf_obs = self._bufs[0]
        self.next_obs = torch.zeros_like(self.buf_obs)
        self.action = torch.zeros(self.num_envs, dtype=torch.int32, device=device)
        self.done = torch.zeros(self.num_envs, device=device)
        self.end_step = torch.zeros(self.num_envs, dtype=torch.int32, device=device)
        self._host_step = 0
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=device)    # (captured rollouts: advance())

    def _kw(self, cur, offset=None):
        return dict(cur_obs=cur, next_obs=self.next_obs, action=self.action, reward=self.reward, terminated=self.terminated,
                    truncated=self.truncated, done=self.done, steps=self.steps, end_step=self.end_step, n=self.num_envs,
                    row_bytes=84 * 84 * 4, A=self.action_space.n, max_steps=self.max_episode_steps, p_term=self.p_term,
                    seed=self.seed, step=self._host_step if offset is None else int(offset),      # eager loops: the host knows
                    step_dev=None if offset is None else self.step_counter)                         # the step index

    def bind_policy_batch(self, batches):
        """batches: two uint8 tensors [2 n, 84 * 84 * 4].  From now on the provider writes its observations straight into the policy's
        input batches -- rows [0, n) of batches[i] are observation buffer i, rows [n, 2 n) receive the next observations of the step
        that makes buffer i current -- so that an on-policy agent's [obs_t ; next_obs_{t-1}] batch of step t IS batches[self._cur]
        (no copies; PPO on frame stacks, agents/ppo_agent.py)."""
        n = self.num_envs
        new = [b[:n].view(n, 84, 84, 4) for b in batches]
        new[self._cur].copy_(self._bufs[self._cur])
        self._bufs, self._nexts = new, [b[n:].view(n, 84, 84, 4) for b in batches]
        self.buf_obs = self._bufs[self._cur]
        self._nexts[self._cur].copy_(self.next_obs)
        self.next_obs = self._nexts[self._cur]

    def reset(self):
        from .. import ops
        ops.synth_frames_step(reset=True, **self._kw(self.buf_obs))
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self, offset=None):
        """offset=t: a step of a captured rollout -- step index = device counter + t, the counter ticks once per rollout through
        advance(T) (as SyntheticMujocoVecEnv); an on-policy agent uses either this or the host-indexed form, never both."""
        from .. import ops
        self._cur ^= 1
        if getattr(self, "_nexts", None) is not None:
            self.next_obs = self._nexts[self._cur]
        ops.synth_frames_step(**self._kw(self._bufs[self._cur], offset))
        self.buf_obs = self._bufs[self._cur]
        if offset is None:
            self._host_step += 1

    def advance(self, k):
        from .. import ops
        ops.counter_add(self.step_counter, int(k))


class SyntheticMujocoVecEnv(_Base):
    """obs (17,) float32, Box(6) actions (HalfCheetah-shaped, configs/ppo/mujoco.yaml); linear-tanh dynamics evaluated by
    one kernel per vector step (xrl_synth_control_step), so that the whole rollout of the continuous-control path can be
    captured into a hipGraph like the CartPole one."""
    graph_safe = True

    def __init__(self, num_envs, seed=1, device="cuda", obs_dim=17, act_dim=6, max_episode_steps=1000):
        super().__init__(num_envs, seed, device, max_episode_steps)
        from .. import ops
        self._ops = ops
        self.seed = int(seed)
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,), np.float32)
        self.action_space = Box(-1.0, 1.0, (act_dim,), np.float32)
        g = torch.Generator().manual_seed(int(seed) + 1000)
        self.A = (torch.randn(obs_dim, obs_dim, generator=g) * 0.3).to(device).contiguous()
        self.B = (torch.randn(act_dim, obs_dim, generator=g) * 0.5).to(device).contiguous()
        self.state = torch.zeros(self.num_envs, obs_dim, device=device)
        self.buf_obs = torch.zeros(self.num_envs, obs_dim, device=device)
        self.next_obs = torch.zeros_like(self.buf_obs)
        self.action = torch.zeros(self.num_envs, act_dim, device=device)
        self.ep_score = torch.zeros(self.num_envs, device=device)
        self.stats = torch.zeros(4, dtype=torch.float64, device=device)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=device)

    def _args(self, offset=0):
        return dict(state=self.state, steps=self.steps, action=self.action, Amat=self.A, Bmat=self.B, obs=self.buf_obs,
                    next_obs=self.next_obs, reward=self.reward, terminated=self.terminated, truncated=self.truncated,
                    ep_score=self.ep_score, stats=self.stats, n=self.num_envs, D=self.obs_dim, A=self.act_dim,
                    max_steps=self.max_episode_steps, seed=self.seed, step=int(offset), step_dev=self.step_counter)

    def reset(self):
        self._ops.synth_control_step(reset=True, **self._args())
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self, offset=None):
        """offset=None: one step, the device counter advances by one.  offset=t (inside a captured rollout whose steps are
        enqueued with static indices): step index = counter + t and the counter is left alone -- call advance(T) once
        after the T steps (one launch per rollout instead of one per step)."""
        if offset is None:
            self._ops.synth_control_step(**self._args())
            self._ops.counter_add(self.step_counter, 1)
        else:
            self._ops.synth_control_step(**self._args(offset))

    def advance(self, k):
        self._ops.counter_add(self.step_counter, int(k))

    def episode_stats(self):
        s = self.stats.cpu().numpy()
        return int(s[0]), float(s[1] / max(s[0], 1.0)), float(s[2] / max(s[0], 1.0))


class SyntheticSMACVecEnv(_Base):
    """SMAC map 3m shape: 3 agents, obs (30,), state (48,), 9 actions with availability masks, 60-step episodes
    (docs/source/documents/benchmark/smac/smac.rst:15,19; obs/state/action dims are SMAC-upstream values).
    One native launch per vector step (xrl_synth_marl_step; Philox streams keyed by seed / env / step).
    `double_buffered`: step_device() leaves the tensors that were buf_obs / buf_state / buf_avail before the call intact
    and rebinds those names to the other of two buffer sets; `prev_steps` = step index of the transition inside its
    episode; `episode_totals` = running totals [episodes finished, env steps in them] (int64, on the device)."""
    graph_safe = False          # buf_* alternate between two tensors: addresses change from step to step
    double_buffered = True

    def __init__(self, num_envs, seed=1, device="cuda", n_agents=3, obs_dim=30, state_dim=48, n_actions=9,
                 max_episode_steps=60, p_term=0.01):
        super().__init__(num_envs, seed, device, max_episode_steps)
        self.n_agents, self.obs_dim, self.state_dim, self.n_actions = n_agents, obs_dim, state_dim, n_actions
        self.seed, self.p_term = int(seed), float(p_term)
        self.agent_keys = [f"agent_{i}" for i in range(n_agents)]
        self.observation_space = {k: Box(-np.inf, np.inf, (obs_dim,), np.float32) for k in self.agent_keys}
        self.action_space = {k: Discrete(n_actions) for k in self.agent_keys}
        self.state_space = Box(-np.inf, np.inf, (state_dim,), np.float32)
        n, N = self.num_envs, n_agents
        self._sets = [(torch.zeros(n, N, obs_dim, device=device), torch.zeros(n, state_dim, device=device),
                       torch.ones(n, N, n_actions, device=device)) for _ in range(2)]
        self._cur = 0
        self.buf_obs, self.buf_state, self.buf_avail = self._sets[0]
        self.agent_mask = torch.ones(n, N, device=device)
        self.action = torch.zeros(n, N, dtype=torch.int32, device=device)
        self.next_obs, self.next_state, self.next_avail = (torch.zeros_like(self.buf_obs), torch.zeros_like(self.buf_state),
                                                           torch.ones_like(self.buf_avail))
        self.rewards = torch.zeros(n, N, device=device)
        self.terminals = torch.zeros(n, N, device=device)
        self.done = torch.zeros(n, device=device)
        self.end_step = torch.zeros(n, dtype=torch.int32, device=device)
        self.prev_steps = torch.zeros(n, dtype=torch.int32, device=device)
        self.episode_totals = torch.zeros(2, dtype=torch.int64, device=device)
        self._host_step = 0

    def _kw(self, new, prev_state=None):
        return dict(buf_obs=new[0], buf_state=new[1], buf_avail=new[2], next_obs=self.next_obs,
                    next_state=self.next_state, next_avail=self.next_avail, action=self.action, rewards=self.rewards,
                    terminals=self.terminals, terminated=self.terminated, truncated=self.truncated, done=self.done,
                    steps=self.steps, end_step=self.end_step, n=self.num_envs, N=self.n_agents, O=self.obs_dim,
                    S=self.state_dim, A=self.n_actions, max_steps=self.max_episode_steps, p_term=self.p_term, seed=self.seed,
                    step=self._host_step, step_dev=None, prev_state=prev_state, prev_steps=self.prev_steps,
                    totals=self.episode_totals)

    def reset(self, counter=None):
        """counter: None (the Philox step index is the host's `_host_step`) or a device counter holding the same value, for
        callers that capture the launch."""
        from .. import ops
        kw = self._kw(self._sets[self._cur])
        if counter is not None:
            kw.update(step=0, step_dev=counter)
        ops.synth_marl_step(reset=True, **kw)
        return self.buf_obs, [{} for _ in range(self.num_envs)]

    def step_device(self):
        self.enqueue_step(self._cur)
        self.flip()

    # -- the two halves of step_device, for callers that capture a vector step into a hipGraph (one graph per buffer set):
    #    enqueue_step(cur) launches the kernel that acts on buffer set `cur` and writes set `cur ^ 1`; with counter != None
    #    the Philox step index comes from that device counter (the caller advances it inside the graph) instead of the
    #    host's `_host_step`.  flip() is the host bookkeeping of one step (rebinding buf_*; no launch).
    def enqueue_step(self, cur, counter=None):
        from .. import ops
        new = self._sets[cur ^ 1]
        kw = self._kw(new, prev_state=self._sets[cur][1])
        if counter is not None:
            kw.update(step=0, step_dev=counter)
        ops.synth_marl_step(**kw)

    def flip(self):
        self._cur ^= 1
        self.buf_obs, self.buf_state, self.buf_avail = self._sets[self._cur]
        self._host_step += 1

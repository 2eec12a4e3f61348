"""DQN agent loop on the HIP engine (xuance/torch/agents/qlearning_family/dqn_agent.py:10-52 with
core/off_policy.py:23-270): greedy action from the eval network, per-env epsilon coin (off_policy.py:138-141), the
reference's epsilon schedule quirk (delta computed with decay_step_greedy / n_envs while current_step grows by n_envs
per vector step, :119-127), replay store, and one learner update per vector step after ``start_training``."""
from argparse import Namespace

import numpy as np
import torch

from .. import ops
from .base import AgentSurface, ActionOutput
from ..learners.dqn_learner import DQN_Learner
from ..memory import HipOffPolicyBuffer, HipOffPolicyBuffer_Atari
from ..nets import DeepQNet
from ..spaces import space2shape


def _get(cfg, name, default=None):
    return getattr(cfg, name, default)


def eps_kstar(start_greedy, end_greedy, delta_egreedy, n_envs):
    """First vector step k whose epsilon start - (k n) delta is <= end_greedy: where OffPolicyAgent._update_explore_factor
    (off_policy.py:119-127) stops updating -- the host constant of xrl_dqn_act_tail_t.eps_sched.  A schedule that never gets
    there (delta <= 0 or NaN, or too small to move the float) returns 0xffffffff."""
    val = lambda k: start_greedy - (k * n_envs) * delta_egreedy      # the reference's own float64 expression
    if not start_greedy > end_greedy:
        return 0
    if not delta_egreedy > 0 or not np.isfinite(delta_egreedy):
        return 0xffffffff
    g = (start_greedy - end_greedy) / (n_envs * delta_egreedy)
    if not np.isfinite(g) or g >= 0xffffffff:
        return 0xffffffff
    k = max(int(np.ceil(g)), 1)
    while k > 1 and val(k - 1) <= end_greedy:                          # (the closed form is within one step of the float64 answer)
        k -= 1
    while val(k) > end_greedy and k < 0xffffffff:
        k += 1
    return k


class DQN_Agent(AgentSurface):
    def __init__(self, config: Namespace, envs, callback=None):
        self.config, self.envs, self.callback = config, envs, callback
        self.device = _get(config, "device", "cuda")
        self._init_surface()
        self.n_envs = envs.num_envs
        self.observation_space, self.action_space = envs.observation_space, envs.action_space
        self.gamma = config.gamma
        self.use_obsnorm, self.use_rewnorm = _get(config, "use_obsnorm", False), _get(config, "use_rewnorm", False)
        self.obsnorm_range, self.rewnorm_range = _get(config, "obsnorm_range", 5.0), _get(config, "rewnorm_range", 5.0)
        self.start_training, self.training_frequency = config.start_training, config.training_frequency
        self.n_epochs = _get(config, "n_epochs", 1)
        self.use_graph_updates = bool(_get(config, "use_hip_graph", True))   # whole update phases as one hipGraph launch
        self.seed = int(_get(config, "seed", 1))
        # dqn_agent.py:28-30
        self.start_greedy, self.end_greedy = config.start_greedy, config.end_greedy
        self.e_greedy = config.start_greedy
        self.delta_egreedy = (self.start_greedy - self.end_greedy) / (config.decay_step_greedy / self.n_envs)
        self.current_step = 0
        dev, n = self.device, self.n_envs
        self.obs_shape = space2shape(self.observation_space)
        self.obs_dim = int(np.prod(self.obs_shape))
        self.atari = _get(config, "env_name", "") == "Atari"
        self.model = self._build_model()
        self.memory = self._build_memory()
        self.learner = self._build_learner(self.config, self.model, self.callback)
        D = self.obs_dim
        self.obs_mean = torch.zeros(D, device=dev)
        self.obs_var = torch.ones(D, device=dev)
        self.obs_count = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
        xdt = torch.uint8 if self.atari else torch.float32
        self.X = torch.zeros(n, D, dtype=xdt, device=dev)      # processed observation
        self.Xn = torch.zeros(n, D, dtype=xdt, device=dev)     # processed next observation
        self.eps_dev = torch.full((1,), float(self.e_greedy), device=dev)
        self._eps_on_device = self.e_greedy
        self._host_step = 0
        self._act_fused = bool(getattr(config, "use_fused_q_tail", True)) and hasattr(self.model, "act_egreedy") and \
            getattr(self.model, "fused_tail", lambda: None)() is not None
        self.act_f = torch.zeros(n, device=dev)
        self.model.plan.ensure(max(n, 2 * config.batch_size))
        assert not (self.atari and self.use_obsnorm), "Atari frames are stored as uint8 (configs/dqn/atari.yaml:42-43)"
        self._started = False
        self._act_calls = 0
        assert not self.use_rewnorm, "reward normalisation is not built for the off-policy loops (every configs/dqn/*.yaml has use_rewnorm: False)"
        # Supplied randomness (replays of recorded runs, set_replay): per vector step the exploration coins [S, n] and the random
        # actions [S, n] (off_policy.py:138-139: torch.rand(n_envs), torch.randint(n_actions)); per update the replay choices
        # [2, batch] = (env_choices, step_choices) of memory_tools.py:376-377.  Both are consumed in order.
        self.explore_tape = None
        self.index_tape = None

    def set_replay(self, coins=None, random_actions=None, indices=None):
        """Replay hook: the loop's random decisions come from the caller (a recorded run) instead of the Philox streams.  With
        coins / random_actions the acting step is the layered one (xrl_egreedy takes supplied draws); with indices every update is
        `learner.update(**memory.sample(indexes))`, launch by launch (the captured update phase draws inside its gather launch:
        tests/test_gpu_offpolicy_agents.py shows the two bit-identical on the same indices)."""
        dev = self.device
        if coins is not None:
            self.explore_tape = (torch.as_tensor(np.asarray(coins, np.float32), device=dev).contiguous(),
                                 torch.as_tensor(np.asarray(random_actions, np.int32), device=dev).contiguous())
            self._act_fused = False
        if indices is not None:
            self.index_tape = [np.asarray(i) for i in indices]          # ([2, batch] int choices; PerDQN_Agent: float64 uniforms)
            self._index_pos = 0

    def _build_model(self):
        c = self.config
        if _get(c, "representation", "Basic_MLP") == "Basic_CNN":
            from ..nets import DeepQCNN
            return DeepQCNN(tuple(self.obs_shape), self.action_space.n, tuple(c.kernels), tuple(c.strides), tuple(c.filters),
                            tuple(c.q_hidden_size), _get(c, "activation", "relu"), device=self.device)
        rep = list(_get(c, "representation_hidden_size", []) or []) if _get(c, "representation", "Basic_MLP") == "Basic_MLP" else []
        return DeepQNet(self.obs_dim, self.action_space.n, rep, list(c.q_hidden_size), _get(c, "activation", "relu"),
                        device=self.device)

    def _build_memory(self):
        c = self.config
        Buffer = HipOffPolicyBuffer_Atari if self.atari else HipOffPolicyBuffer
        return Buffer(self.observation_space, self.action_space, None, self.n_envs, c.buffer_size, c.batch_size,
                      device=self.device)

    def _build_learner(self, *args):
        return DQN_Learner(*args)

    def _update_explore_factor(self):                          # off_policy.py:119-127
        if self.e_greedy is not None:
            if self.e_greedy > self.end_greedy:
                self.e_greedy = self.start_greedy - self.current_step * self.delta_egreedy

    def _eps_tensor(self):
        """epsilon in device memory, brought up to date when somebody wants it there: the training loop hands the value to its
        acting launch as an argument (a fill launch per vector step was 4.7 us of the 170 us DQN-C3 step)."""
        if self.e_greedy != self._eps_on_device:
            self.eps_dev.fill_(float(self.e_greedy))
            self._eps_on_device = self.e_greedy
        return self.eps_dev

    def _normalize(self, raw, out, update):
        n, D = self.n_envs, self.obs_dim
        if self.atari:
            out.copy_(raw.reshape(n, D))
            return
        ops.obs_normalize(x=raw.reshape(n, D), mean=self.obs_mean, var=self.obs_var, count=self.obs_count, out0=out,
                          out1=None, n=n, D=D, ld_x=D, ld0=D, ld1=D, update=int(update and self.use_obsnorm),
                          normalize=int(self.use_obsnorm), range=float(self.obsnorm_range))

    def train(self, train_steps):
        env = self.envs
        if not self._started:
            env.reset()
            self._started = True
        info = {}
        left = train_steps
        while left > 0:
            if left >= 2 and self._pair_ready():
                self._run_pair()
                left -= 2
                continue
            info = self._eager_step(train_steps, info)
            left -= 1
        if self.use_graph_updates and hasattr(self.learner, "flush_info"):
            info = dict(self.learner.flush_info() or info)      # the one host read of this call (phases ran unsynchronised)
        if hasattr(env, "episode_stats"):
            eps, score, length = env.episode_stats()
            info.update({"episodes": eps, "mean_episode_score": score, "mean_episode_length": length})
        info["epsilon"] = self.e_greedy
        return info

    def _eager_step(self, train_steps, info):
        """One vector step, launch by launch (off_policy.py:207-269)."""
        env, n, A = self.envs, self.n_envs, self.action_space.n
        zero_copy = self.atari and getattr(env, "double_buffered", False)     # uint8 frames go to the ring as they are
        shp = (n,) + tuple(self.obs_shape)
        if zero_copy:
            X = env.buf_obs.view(n, -1)                   # stays intact over step_device(): the env alternates buffers
        else:
            self._normalize(env.buf_obs if self.atari else env.buf_obs.float(), self.X, update=True)   # obs_rms.update; process
            X = self.X
        if self._act_fused:                                    # (convolutional Q network: pool .. epsilon-greedy in one launch)
            self.model.act_egreedy(X[:n], n, None, env.action, self.act_f, self.seed, self._host_step, eps=float(self.e_greedy))
        else:
            q = self.model.forward(X[:n], n)
            tape = {}
            if self.explore_tape is not None:
                tape = dict(uniforms=self.explore_tape[0][self._host_step], randoms=self.explore_tape[1][self._host_step])
            ops.egreedy(q=q, eps_dev=None, eps=float(self.e_greedy), action=env.action, action_f=self.act_f, n=n, A=A, ld=q.stride(0), seed=self.seed,
                        step=self._host_step, step_dev=None, **tape)   # eager loop: the host knows the step index
        env.step_device()
        self._host_step += 1
        if zero_copy:
            Xn = env.next_obs.view(n, -1)
        else:
            self._normalize(env.next_obs if self.atari else env.next_obs.float(), self.Xn, update=False)
            Xn = self.Xn
        # off_policy.py:221-224 (device tensors: nothing is copied or synchronised unless the callback reads them)
        self._cb("on_train_step", self.current_step, envs=env, model=self.model, obs=X.view(shp), acts=self.act_f,
                 next_obs=Xn.view(shp), rewards=env.reward, terminals=env.terminated, truncations=getattr(env, "truncated", None),
                 infos=None, train_steps=train_steps)
        self.memory.store(X.view(shp), self.act_f, env.reward, env.terminated, Xn.view(shp))
        if self.current_step > self.start_training and self.current_step % self.training_frequency == 0:
            info = self._train_epochs(train_steps) or info
            self._cb("on_train_epochs_end", self.current_step, model=self.model, memory=self.memory, train_steps=train_steps,
                     update_info=info)                          # :232-234
        self.current_step += n
        self._update_explore_factor()
        self._cb("on_train_step_end", self.current_step, envs=env, model=self.model, train_steps=train_steps, train_info=info)   # :268-269
        return info

    # -- two vector steps as ONE graph launch (acting, provider, store, update phase -- twice: the provider's two observation
    #    buffers alternate, so the addresses repeat with period 2).  Nothing in it takes an argument that changes from step to step:
    #    the step index is a device counter (the provider's), epsilon is computed from it by the acting launch (the host's float64
    #    arithmetic), the ring slot and the filled-slot count by the store launch, the draw counter and the loss sums ride in the
    #    optimiser launch (DQN_Learner.update_from_buffer).  The host mirrors (current_step, e_greedy, memory.ptr / size, the step
    #    counters) advance without reading anything back, so the launch-by-launch loop can take over at any pair boundary.
    def _eps_kstar(self):
        if getattr(self, "_kstar", None) is None:
            self._kstar = eps_kstar(self.start_greedy, self.end_greedy, self.delta_egreedy, self.n_envs)
        return self._kstar

    def _pair_ready(self):
        env, lr, n = self.envs, self.learner, self.n_envs
        if self.explore_tape is not None or self.index_tape is not None:
            return False
        if not (bool(_get(self.config, "use_step_graph", True)) and self.use_graph_updates and self._act_fused and self.atari
                and getattr(env, "double_buffered", False) and getattr(env, "graph_safe_even", False)
                and hasattr(lr, "phase_ready") and type(self.memory) in (HipOffPolicyBuffer, HipOffPolicyBuffer_Atari)):
            return False
        if any(self._has_cb(h) for h in ("on_train_step", "on_train_epochs_end", "on_train_step_end")) or lr.needs_collective():
            return False
        from ..learners.base import _NullCallback
        if not isinstance(lr.callback, _NullCallback) or not lr.phase_ready(self.memory, self.n_epochs):
            return False
        if self.e_greedy is None or self.current_step != self._host_step * n:
            return False
        for cs in (self.current_step, self.current_step + n):
            if not (cs > self.start_training and cs % self.training_frequency == 0):
                return False
        par = getattr(self, "_pair_parity", None)
        return par is None or par == env._cur                      # (the other parity: one launch-by-launch step first)

    def _enqueue_pair(self, d_act, bias):
        env, n, mem = self.envs, self.n_envs, self.memory
        shp = (n,) + tuple(self.obs_shape)
        sched = (n, self._eps_kstar(), self.start_greedy, self.delta_egreedy)
        for t in (0, 1):
            X = env.buf_obs.view(n, -1)
            self.model.act_egreedy(X, n, None, env.action, self.act_f, self.seed, d_act + t, step_dev=env.step_counter, eps_sched=sched)
            env.step_device(offset=t)
            mem.store_ring(X.view(shp), self.act_f, env.reward, env.terminated, env.next_obs, env.step_counter, t, bias, mirror=False)
            self.learner.enqueue_phase()
        env.advance(2)

    def _run_pair(self):
        env, mem, lr, n = self.envs, self.memory, self.learner, self.n_envs
        if getattr(self, "_vc_mirror", None) != env._host_step:     # the device counter follows the provider's own step index
            env.step_counter.fill_(int(env._host_step))
            self._vc_mirror = env._host_step
        d_act = (self._host_step - env._host_step) & 0xffffffff
        # (+ the workspaces' signature: a get_actions / test call on more rows reallocates the plan's activations and the convolution
        #  workspace the captured launches point into)
        key = (d_act, mem.ring_bias(env._host_step), env._cur, id(mem), self.model.plan.cap,
               getattr(getattr(self.model, "conv", None), "ws_gen", 0))
        if getattr(self, "_pair_key", None) != key:
            torch.cuda.synchronize()
            g = ops.Graph()
            with g:
                self._enqueue_pair(d_act, key[1])
            self._pair_graph, self._pair_key, self._pair_parity = g, key, env._cur
        lr.ensure_live_images()
        self._pair_graph.launch()
        for _ in range(2):
            mem.advance_mirrors(1)
            self._host_step += 1
            env._host_step += 1
            self.current_step += n
            self._update_explore_factor()
        self._vc_mirror += 2
        lr.note_phases(mem, self.n_epochs, 2)

    def _train_epochs(self, train_steps):
        if self.index_tape is not None:
            info = {}
            for _e in range(self.n_epochs):
                env_c, step_c = self.index_tape[self._index_pos].astype(np.int64)
                self._index_pos += 1
                info = self.learner.update(**self.memory.sample(indexes=env_c * self.memory.n_size + step_c))
            return info
        if self.use_graph_updates:
            return self.learner.update_from_buffer(self.memory, self.n_epochs, seed=self.seed, sync=False)
        info = {}
        for _e in range(self.n_epochs):
            info = self.learner.update(**self.memory.sample())
        return info

    # -- acting outside the training loop (core/off_policy.py:150-171, 272-350) ---------------------------------------------
    @torch.no_grad()
    def get_actions(self, observations, test_mode=False):
        """OffPolicyAgent.get_actions: PROCESSED observations [m, *obs_shape] -> ActionOutput(env_actions [m] int64):
        greedy actions of the eval network; unless test_mode, the per-env epsilon coin on top (exploration, :129-148)."""
        dev = self.model.params.device
        X = torch.as_tensor(np.asarray(observations) if not isinstance(observations, torch.Tensor) else observations, device=dev)
        X = X.reshape(-1, self.obs_dim).to(torch.uint8 if self.atari else torch.float32).contiguous()
        m, A = X.shape[0], self.action_space.n
        q = self.model.forward(X, m)
        act = torch.zeros(m, dtype=torch.int32, device=dev)
        eps = self._eps_tensor() if not test_mode else torch.zeros(1, device=dev)
        ops.egreedy(q=q, eps_dev=eps, action=act, action_f=None, n=m, A=A, ld=q.stride(0), seed=self.seed,
                    step=(1 << 20) + self._act_calls, step_dev=None)
        self._act_calls = (self._act_calls + 1) & 0xfffff
        return ActionOutput(env_actions=act.cpu().numpy().astype(np.int64), values=None, distributions=None, log_probs=None)

    def _test_actions(self, obs, deterministic=True):
        # off_policy.py:319-321: obs_rms.update(obs); obs = _process_observation(obs); get_actions(obs, test_mode=True)
        dev, m = self.model.params.device, len(obs)
        raw = torch.as_tensor(np.asarray(obs), device=dev).reshape(m, self.obs_dim)
        if self.atari or not self.use_obsnorm:
            return self.get_actions(raw, test_mode=True).env_actions
        X = torch.empty(m, self.obs_dim, device=dev)
        ops.obs_normalize(x=raw.float().contiguous(), mean=self.obs_mean, var=self.obs_var, count=self.obs_count, out0=X, out1=None,
                          n=m, D=self.obs_dim, ld_x=self.obs_dim, ld0=self.obs_dim, ld1=self.obs_dim, update=1, normalize=1,
                          range=float(self.obsnorm_range))
        return self.get_actions(X, test_mode=True).env_actions


class DDQN_Agent(DQN_Agent):
    """Double DQN (xuance/torch/agents/qlearning_family/ddqn_agent.py:10-33): DQN_Agent whose learner takes the target
    action from the eval network (DDQN_Learner; `learner: "DDQN_Learner"` in configs/ddqn/*.yaml)."""

    def _build_learner(self, *args):
        from ..learners.dqn_learner import DDQN_Learner
        return DDQN_Learner(*args)


class DuelDQN_Agent(DQN_Agent):
    """Dueling DQN (xuance/torch/agents/qlearning_family/dueldqn_agent.py:12-48): DuelingDeepQNetwork (value + advantage
    streams on the shared representation, deep_q_network.py:102-171) with DuelDQN_Learner."""

    def _build_model(self):
        c = self.config
        assert _get(c, "representation", "Basic_MLP") == "Basic_MLP", "the dueling head is built for MLP representations"
        return DeepQNet(self.obs_dim, self.action_space.n, list(_get(c, "representation_hidden_size", []) or []),
                        list(c.q_hidden_size), _get(c, "activation", "relu"), device=self.device, dueling=True)

    def _build_learner(self, *args):
        from ..learners.dqn_learner import DuelDQN_Learner
        return DuelDQN_Learner(*args)


class PerDQN_Agent(DQN_Agent):
    """DQN with prioritized replay (xuance/torch/agents/qlearning_family/perdqn_agent.py:12-96): HipPerOffPolicyBuffer,
    PerDQN_Learner, `sample(beta) -> update -> update_priorities` per epoch (:43-49) and the linear beta schedule
    PER_beta += (1 - PER_beta0) / train_steps after every training step (:72).  |td| never leaves the device."""

    def __init__(self, config, envs, callback=None):
        self.PER_beta0 = self.PER_beta = float(config.PER_beta0)
        super().__init__(config, envs, callback)

    def _build_memory(self):
        from ..memory import HipPerOffPolicyBuffer
        c = self.config
        return HipPerOffPolicyBuffer(self.observation_space, self.action_space, None, self.n_envs, c.buffer_size,
                                     c.batch_size, alpha=c.PER_alpha, device=self.device,
                                     obs_dtype=torch.uint8 if self.atari else torch.float32)

    def _build_learner(self, *args):
        from ..learners.dqn_learner import PerDQN_Learner
        return PerDQN_Learner(*args)

    def _update_explore_factor(self):
        # perdqn_agent.py:104-105: this agent's OWN rule -- a fixed decrement per vector step while above end_greedy (DQN_Agent
        # recomputes start - current_step * delta, off_policy.py:119-127); pinned by tests/golden/agent_perdqn.npz
        if self.e_greedy > self.end_greedy:
            self.e_greedy -= self.delta_egreedy

    def _pair_ready(self):
        return False                                        # (the captured vector-step pair evaluates DQN_Agent's schedule)

    def _train_epochs(self, train_steps):
        info = {}
        for _e in range(self.n_epochs):
            uni = None
            if self.index_tape is not None:                 # replay: the proportional draws' uniforms [n_envs, batch / n_envs] (set_replay)
                uni = self.index_tape[self._index_pos]
                self._index_pos += 1
            samples = self.memory.sample(self.PER_beta, uniforms=uni)
            td, info = self.learner.update(**samples)
            self.memory.update_priorities(samples["step_choices"], td)
        self.PER_beta += (1 - self.PER_beta0) / train_steps
        return info

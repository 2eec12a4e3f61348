"""QMIX agents loop on the HIP engine, feed-forward agents with parameter sharing
(xuance/torch/agents/multi_agent_rl/qmix_agents.py:12-93 with core/off_policy_marl.py:90-424 and
base/agents_marl.py:339-376): per vector step the [n_envs, n_agents, obs] batch goes through the shared Q-network,
actions are the masked greedy ones unless the step's single exploration coin lands (off_policy_marl.py:236-243),
transitions are stored and, once ``current_step >= start_training`` and on the training frequency, ``n_epochs``
updates are made (off_policy_marl.py:376-377).

Recurrent agents (``use_rnn``, the default of configs/qmix/sc2/3m.yaml): training alternates ``run_episodes(n_envs)`` and
``n_epochs`` updates (off_policy_marl.py:335-349); the GRU state of every (env, agent) row lives on the device, is
carried by xrl_gru_forward between steps and zeroed for the rows of a finished env (:504-505); steps go into the
staging rows of HipMARLOffPolicyBufferRNN and finished episodes into its ring without the host knowing which envs
finished -- the one host read per vector step is the (episodes, env-steps) count that run_episodes' loop condition and
the exploration schedule need (:494,532-534)."""
from argparse import Namespace

import numpy as np
import torch

from .. import ops
from .base import AgentSurface
from ..learners.qmix_learner import QMIX_Learner, VDN_Learner, IQL_Learner
from ..memory_marl import HipMARLOffPolicyBuffer, HipMARLOffPolicyBufferRNN
from ..nets import MixingQNet


def _get(cfg, name, default=None):
    return getattr(cfg, name, default)


class QMIX_Agents(AgentSurface):
    mixer_name, learner_cls = "QMIX", QMIX_Learner
    eps_decay_per_env = True        # qmix_agents.py:40, vdn_agents.py:38: delta = (start - end) / (decay_step_greedy / n_envs);
    #                                 iql_agents.py:37: / decay_step_greedy

    def __init__(self, config: Namespace, envs, callback=None):
        self.config, self.envs, self.callback = config, envs, callback
        self.device = _get(config, "device", "cuda")
        self._init_surface()
        self.n_envs = envs.num_envs
        self.agent_keys = list(envs.agent_keys)
        self.n_agents = len(self.agent_keys)
        self.use_actions_mask = _get(config, "use_actions_mask", True)
        self.use_rnn = bool(_get(config, "use_rnn", False))
        self.start_training, self.training_frequency = config.start_training, config.training_frequency
        self.n_epochs = _get(config, "n_epochs", 1)
        self.use_graph_updates = bool(_get(config, "use_hip_graph", True))   # whole update phases as one hipGraph launch
        self.seed = int(_get(config, "seed", 1))
        self.start_greedy, self.end_greedy = config.start_greedy, config.end_greedy
        self.e_greedy = config.start_greedy
        self.delta_egreedy = (self.start_greedy - self.end_greedy) / \
            ((config.decay_step_greedy / self.n_envs) if self.eps_decay_per_env else config.decay_step_greedy)
        self.current_step = 0
        k0 = self.agent_keys[0]
        self.obs_dim = envs.observation_space[k0].shape[0]
        self.n_actions = envs.action_space[k0].n
        self.state_dim = envs.state_space.shape[0]
        self.model = self._build_model()
        self.memory = self._build_memory()
        self.learner = self._build_learner(self.config, self.agent_keys, self.model, self.callback)
        dev, R = self.device, self.n_envs * self.n_agents
        self.eps_dev = torch.full((1,), float(self.e_greedy), device=dev)
        self._host_step = 0
        self.episode_loop_lag = max(0, int(getattr(config, "episode_loop_lag", 1)))   # graph launches enqueued ahead of the host
        self.episode_loop_unroll = max(2, int(getattr(config, "episode_loop_unroll", 4)) // 2 * 2)   # vector steps per graph launch
        self.act_f = torch.zeros(self.n_envs, self.n_agents, device=dev)
        # The reference's loops store, for the vector step after ANY episode end, the reset state of the last finished env in every
        # env's row (off_policy_marl.py:395, :507 assign `state = info[i]["reset_state"]`, the buffer broadcasts the single vector:
        # xrl_marl_stored_state).  On by default -- buffers and updates then equal the reference's on the same simulator outputs
        # (tests/test_gpu_agent_replay.py); `reference_state_broadcast: False` stores every env's own state.
        self.state_broadcast = bool(_get(config, "reference_state_broadcast", True))
        if self.state_broadcast and not hasattr(envs, "done"):
            #  a provider without per-env `done` flags of the last vector step (envs.done, as envs/synthetic.py has them)
            # cannot feed the reproduction of that defect: store every env's own state and say so
            import warnings
            warnings.warn("xuance_amd: reference_state_broadcast needs the provider's per-env `done` flags (envs.done); this provider "
                          "has none -- storing every env's own global state (reference_state_broadcast: False)")
            self.state_broadcast = False
        self._stored_state = torch.zeros(self.n_envs, self.state_dim, device=dev) if self.state_broadcast else None
        # ... and zero the recurrent state of flattened row i -- not of env i's rows -- when env i finishes (init_rnn_states_item is
        # handed batch_index = [i_env] for a state whose batch axis is n_envs * n_agents: value_factorization.py:161-167 with
        # representations/rnn.py:86-92).  `reference_rnn_reset: False` resets the finished env's own rows.
        self.reference_rnn_reset = bool(_get(config, "reference_rnn_reset", True))
        # Supplied randomness (replays of recorded runs, set_replay): per vector step the exploration coin [S] and the uniforms
        # [S, n_envs * n_agents] behind the random available actions; per update the replay choices.  Consumed in order.
        self.explore_tape = None
        self.index_tape = None
        if self.use_rnn:
            self.rnn_h = torch.zeros(R, self.model.RH, device=dev)           # init_rnn_states (value_factorization.py:151-159)
            self.rnn_c = torch.zeros(R, self.model.RH, device=dev) if self.model.lstm else None   # LSTM cell states (rnn.py:79-84)
            self.reset_rows = torch.zeros(R, device=dev)
            self._ends = torch.zeros(self.n_envs, device=dev)
            self._ends_h = torch.zeros(self.n_envs).pin_memory() if torch.cuda.is_available() else torch.zeros(self.n_envs)
            self._totals_h = torch.zeros(2, dtype=torch.int64)
            if torch.cuda.is_available():
                self._totals_h = self._totals_h.pin_memory()
        else:
            self.model.agent_plan.ensure(max(R, 2 * config.batch_size * self.n_agents))
        self._started = False

    def _build_model(self):
        c = self.config
        return MixingQNet(self.n_agents, self.obs_dim, self.n_actions, self.state_dim,
                          list(_get(c, "representation_hidden_size", [64])), list(_get(c, "q_hidden_size", [64])),
                          _get(c, "hidden_dim_mixing_net", 32), _get(c, "hidden_dim_hyper_net", 32),
                          _get(c, "activation", "relu"), device=self.device, use_rnn=self.use_rnn,
                          fc_hidden=list(_get(c, "fc_hidden_sizes", [64])), recurrent_hidden=_get(c, "recurrent_hidden_size", 64),
                          mixer=self.mixer_name, rnn=_get(c, "rnn", "GRU"))

    def _build_memory(self):
        c, env = self.config, self.envs
        if self.use_rnn:                                       # off_policy_marl.py:106
            return HipMARLOffPolicyBufferRNN(self.agent_keys, env.state_space, env.observation_space, env.action_space,
                                             self.n_envs, c.buffer_size, c.batch_size, env.max_episode_steps,
                                             device=self.device, use_actions_mask=self.use_actions_mask,
                                             avail_actions_shape={k: (self.n_actions,) for k in self.agent_keys})
        return HipMARLOffPolicyBuffer(self.agent_keys, env.state_space, env.observation_space, env.action_space,
                                      self.n_envs, c.buffer_size, c.batch_size, device=self.device,
                                      use_actions_mask=self.use_actions_mask,
                                      avail_actions_shape={k: (self.n_actions,) for k in self.agent_keys})

    def _build_learner(self, *args):
        return self.learner_cls(*args)

    def _update_explore_factor(self, push=True):               # off_policy_marl.py:197-204
        if self.e_greedy > self.end_greedy:
            self.e_greedy = self.start_greedy - self.delta_egreedy * self.current_step
        else:
            self.e_greedy = self.end_greedy
        if push:
            self._push_eps()

    def _push_eps(self):
        """epsilon into device memory for the launches that read it there (the feed-forward loop hands the value to its acting
        launch as an argument instead: no fill launch per vector step)."""
        if self.e_greedy != getattr(self, "_eps_on_device", None):
            self.eps_dev.fill_(float(self.e_greedy))
            self._eps_on_device = self.e_greedy

    def run_episodes(self, n_episodes):                        # off_policy_marl.py:426-546 (training mode)
        env, n, N, A, mem = self.envs, self.n_envs, self.n_agents, self.n_actions, self.memory
        R = n * N
        fused_act = bool(getattr(self.config, "use_fused_acting", True))
        self._refresh_act_image(fused_act)
        # an env that alternates its observation buffers and keeps running episode totals saves the copies and the
        # reductions of a step (envs/synthetic.py); any other env goes through clones and two small sums
        two_buf, totals = getattr(env, "double_buffered", False), getattr(env, "episode_totals", None)
        # (the captured loop draws from the Philox streams -- a supplied exploration tape needs the eager loop)
        if self.use_graph_updates and two_buf and totals is not None and hasattr(env, "enqueue_step") and self.explore_tape is None:
            return self._run_episodes_captured(n_episodes, totals)
        self._call_prologue(env, mem)
        episodes = 0
        first = True
        while episodes < n_episodes:
            if two_buf:
                obs, state, avail = env.buf_obs, env.buf_state, env.buf_avail
            else:
                obs, state, avail = env.buf_obs.clone(), env.buf_state.clone(), env.buf_avail.clone()
                steps = env.steps.clone()
            if self.state_broadcast:                            # what the reference stores as this step's state (see __init__;
                ops.marl_stored_state(state, None if first else env.done, self._stored_state)   # a call starts from the reset state)
                state, first = self._stored_state, False
            self.model.act_step(obs.view(R, -1), R, self.rnn_h, self.reset_rows, self.rnn_c, fused=fused_act and self.explore_tape is None,
                                select=dict(avail=avail if self.use_actions_mask else None, eps_dev=self.eps_dev, action=env.action,
                                            action_f=self.act_f, seed=self.seed, step=self._host_step,
                                            step_dev=None, **self._tape_kw()))    # eager loop: the host knows the step index
            env.step_device()
            self._host_step += 1
            mem.store(obs=obs, actions=self.act_f, rewards=env.rewards, terminals=env.terminals, agent_mask=env.agent_mask,
                      avail_actions=avail, state=state, episode_steps=env.prev_steps if two_buf else steps)
            mem.finish_paths(env.done, env.end_step, obs=env.next_obs, state=env.next_state, avail_actions=env.next_avail)
            self._set_reset_rows(self.reset_rows, env.done, n, N)
            # the step's only host read: episode lengths of the envs that finished, in env order -- `current_step +=
            # info[i]["episode_step"]` and the epsilon update once per finished env (:532-534)
            torch.mul(env.done, env.end_step, out=self._ends)
            self._ends_h.copy_(self._ends)
            for v in self._ends_h.tolist():
                if v > 0:
                    episodes += 1
                    self.current_step += int(v)
                    self._update_explore_factor(push=False)
            self._push_eps()

    def _set_reset_rows(self, rows, done, m, N):
        """rows [m * N] <- 1 where the recurrent state restarts from zero after this step (see reference_rnn_reset)."""
        if self.reference_rnn_reset:
            rows.zero_()
            rows[:m].copy_(done)
        else:
            rows.view(m, N).copy_(done[:, None].expand(m, N))

    def _refresh_act_image(self, fused_act=True):
        """The acting launch's weight image follows the parameters through the optimiser launch's mirrors; rebuild it (two
        launches) when anything else wrote them: the model's `version` moved (load_state_dict, copy_target), or a learner
        path without mirrors / a checkpoint load raised `_act_stale`."""
        m = self.model
        if fused_act and m.act_image() is not None and \
                (getattr(m, "_act_stale", True) or getattr(m, "version", 0) != getattr(m, "_act_version", -1)):
            m.act_image().refresh()
            m._act_stale, m._act_version = False, getattr(m, "version", 0)

    def _call_prologue(self, env, mem, counter=None):
        """What a run_episodes call starts with (:436-462): env reset, empty staging rows, zero recurrent state and reset flags."""
        env.reset() if counter is None else env.reset(counter=counter)
        mem.clear_episodes()
        self.rnn_h.zero_()
        if self.rnn_c is not None:
            self.rnn_c.zero_()
        self.reset_rows.zero_()
        if self.state_broadcast and hasattr(env, "done"):
            env.done.zero_()                                    # (no episode end precedes a call's first step: xrl_marl_stored_state)

    def _run_episodes_captured(self, n_episodes, totals):
        """run_episodes with the vector step -- acting forward incl. the recurrence and the action selection, provider step,
        staging store, episode close, and (marl_loop_gate) reset flags, RNG counters, ring pointers and the loop's own
        bookkeeping: episodes finished, current_step, the e-greedy schedule, `episodes < n_episodes` -- captured, and
        `episode_loop_unroll` (K, even: the observation-buffer sets alternate) consecutive steps per graph launch: at 5
        launches and ~35 us of device time per step, one hipGraphLaunch per step leaves the loop host-bound.  The host
        enqueues graph j + `episode_loop_lag` before it reads whether the step after graph j still belongs to the call (one
        4-byte pinned read per graph, never waited on while the device has work), so a call ends with up to
        K * (lag + 1) - 1 dry steps that change nothing the eager loop would see (csrc/episodes.hip)."""
        env, lag, K = self.envs, self.episode_loop_lag, self.episode_loop_unroll
        assert env._cur == 0, "the captured steps start on observation-buffer set 0"
        graphs = self._steps_graph()
        g, ring = self._gate, self._flag_h.numel()
        # per-call state of the gate: one host->device copy of a pinned template (call = [current_step, n_episodes], snap = 0,
        # e_state, active = 1, seq = 0, active_f = 1) + the provider's totals as the call's base
        t = self._gate_tpl
        t["call"][0], t["call"][1] = self.current_step, n_episodes
        t["e_state"][0] = self.e_greedy
        self._gate_buf.copy_(self._gate_tpl_raw, non_blocking=True)
        self._prologue_g.launch()                                          # the call's prologue + base <- totals: one launch
        flags = self._flag_np
        flags[:] = 0                                                       # 0 = "this step's gate has not run yet"
        launched, over, spins = 0, False, 0
        while not over:
            graphs[launched % len(graphs)].launch()                        # (alternating executables: see _steps_graph)
            launched += 1
            j = launched - 1 - lag                                         # the newest graph whose outcome is read now
            if j >= 0:
                slot = (j * K + K - 1) % ring                              # its last step's flag: polled in pinned memory -- an
                while flags[slot] == 0:                                    # event record + synchronize per graph cost 40 us of
                    spins += 1                                             # device time each (108 -> 150 us per graph)
                    if spins > 200_000_000:
                        raise ops.XrlError("run_episodes: the captured vector steps did not report back")
                over = int(flags[slot]) == 1                               # 1: "the step after graph j is not part of the call"
                flags[slot] = 0
        self._gate_out_h.copy_(self._gate_all)                             # ONE read of the call's outcome: gate block + RNG counters
        out = self._gate_out
        self._host_step, env._host_step = int(out["rng"][0]), int(out["rng"][1])   # (advanced by the steps that counted)
        self.current_step += int(out["snap"][1])
        self.e_greedy = self._eps_on_device = float(out["e_state"][0])

    def _steps_graph(self):
        """K captured vector steps of run_episodes, alternating between the provider's two buffer sets (same launches, same
        order and same Philox step indices as the eager loop: the indices come from device counters that start at the host's
        values)."""
        if getattr(self, "_steps_g", None) is not None:
            return self._steps_g
        dev, K, lag = self.device, self.episode_loop_unroll, self.episode_loop_lag
        env, n, N, A, mem = self.envs, self.n_envs, self.n_agents, self.n_actions, self.memory
        R = n * N
        self._gate_all = torch.zeros(72, dtype=torch.uint8, device=dev)              # [0, 64): gate state, [64, 72): RNG counters
        self._rng_dev = self._gate_all[64:72].view(torch.int32)                      # [agent step, provider step]
        self._rng_dev.copy_(torch.tensor([self._host_step, env._host_step], dtype=torch.int32))
        # gate state: one 64-byte device block with typed views (+ `base` apart: it is copied from the provider's totals),
        # initialised per call from a pinned template of the same layout
        def carve(raw):
            return dict(call=raw[0:16].view(torch.int64), snap=raw[16:32].view(torch.int64), e_state=raw[32:40].view(torch.float64),
                        active=raw[40:44].view(torch.int32), seq=raw[44:48].view(torch.int32), active_f=raw[48:52].view(torch.float32))
        self._gate_buf = self._gate_all[:64]
        self._gate_tpl_raw = torch.zeros(64, dtype=torch.uint8).pin_memory()
        self._gate_tpl = carve(self._gate_tpl_raw)
        self._gate_tpl["active"][0] = 1
        self._gate_tpl["active_f"][0] = 1.0
        self._gate = gt = dict(base=torch.zeros(2, dtype=torch.int64, device=dev), **carve(self._gate_buf))
        self._gate_out_h = torch.zeros(72, dtype=torch.uint8).pin_memory()
        self._gate_out = dict(rng=self._gate_out_h[64:72].view(torch.int32), **carve(self._gate_out_h))
        ring = K * (lag + 2)                                                          # per-step flags in pinned memory
        self._flag_h = torch.zeros(ring, dtype=torch.int32).pin_memory()
        self._flag_np = self._flag_h.numpy()
        gate_const = dict(host_flags=ops.host_device_pointer(self._flag_h), ring=ring)
        fused = bool(getattr(self.config, "use_fused_acting", True))
        self.model.seq_workspace(2, R, 1)                                             # (no allocation inside the capture)
        if self.model.act_image() is not None:
            self.model.act_q_buffer(R)
        torch.cuda.synchronize()
        # lag + 2 executables of the same capture, launched in turn: a launch of an executable whose previous launch is still
        # running waits for it on the HOST (measured: one executable = a launch every 150 us for 108 us of device work, no
        # overlap at all), so consecutive launches must be different executables for the host to run ahead
        self._prologue_g = ops.Graph()                                             # (~270 us of host time per call as eager calls)
        self._state_in_gate = bool(getattr(self.config, "state_broadcast_in_gate", True))
        with self._prologue_g:
            self._call_prologue(env, mem, counter=self._rng_dev[1:2])
            if self.state_broadcast and self._state_in_gate:
                ops.marl_stored_state(env._sets[0][1], None, self._stored_state)   # the first step of a call stores the reset state
            torch.add(env.episode_totals, 0, out=gt["base"])                       # (a kernel, not a memcpy node)

        def capture():
            g = ops.Graph()
            with g:
                for k in range(K):
                    cur = k & 1
                    obs, state, avail = env._sets[cur]
                    fold = {}
                    if self.state_broadcast:                     # what the reference stores as this step's state (see __init__):
                        if self._state_in_gate:                  # produced by the PREVIOUS step's gate launch (the call's first: prologue)
                            fold = dict(next_state=env._sets[cur ^ 1][1], stored_state=self._stored_state, state_dim=self.state_dim)
                        else:
                            ops.marl_stored_state(state, env.done, self._stored_state)
                        state = self._stored_state
                    self.model.act_step(obs.view(R, -1), R, self.rnn_h, self.reset_rows, self.rnn_c, fused=fused,
                                        select=dict(avail=avail if self.use_actions_mask else None, eps_dev=self.eps_dev,
                                                    action=env.action, action_f=self.act_f, seed=self.seed, step=0,
                                                    step_dev=self._rng_dev[0:1]))
                    env.enqueue_step(cur, counter=self._rng_dev[1:2])
                    # reset flags of the finished envs' rows, RNG counters (+active), ring pointers and the loop's bookkeeping
                    gate_kw = dict(totals=env.episode_totals, start_greedy=float(self.start_greedy),
                                   end_greedy=float(self.end_greedy), delta_greedy=float(self.delta_egreedy),
                                   eps_dev=self.eps_dev, done=env.done, reset_rows=self.reset_rows, counters=self._rng_dev,
                                   n_envs=n, n_agents=N, ptr_size=mem.ptr_size, buffer_size=mem.buffer_size,
                                   reset_rule=int(self.reference_rnn_reset), end_step=env.end_step, **fold, **gate_const, **gt)
                    # (round 6, config.gate_in_finish: True -- the gate rides in the store + finish launch, whose last block carries it
                    #  (xrl_episode_store_finish_gate).  Same results, measured SLOWER: 576 blocks drawing a ticket from one word cost
                    #  more than the launch they save, loop 1.10 M vs 1.22 M env-steps/s; off by default)
                    merged = bool(getattr(self.config, "gate_in_finish", False))
                    mem.store_and_finish(dict(obs=obs, actions=self.act_f, rewards=env.rewards, terminals=env.terminals,
                                              agent_mask=env.agent_mask, avail_actions=avail, state=state),
                                         env.prev_steps, env.done, env.end_step, obs=env.next_obs, state=env.next_state,
                                         avail_actions=env.next_avail, gate=gt["active_f"],   # a dry step closes no episode
                                         loop_gate=gate_kw if merged else None)
                    if not merged:
                        ops.marl_loop_gate(**gate_kw)
            return g
        self._steps_g = [capture() for _ in range(lag + 2)]
        return self._steps_g

    def set_replay(self, coins=None, uniforms=None, indices=None):
        """Replay hook: the loop's random decisions come from the caller (a recorded run) instead of the Philox streams -- coins [S]
        (ONE per vector step, off_policy_marl.py:236) and uniforms [S, n_envs * n_agents] (row r takes its floor(u n_avail)-th
        available action when the coin lands) make the acting step the layered one (xrl_marl_select_actions takes supplied draws);
        with indices ([2, batch] = (env_choices, step_choices) of memory_tools_marl.py:755-756 per update, or [batch] episode rows for
        the recurrent buffer) every update is `learner.update(memory.sample(indexes))`, launch by launch."""
        dev = self.device
        if coins is not None:
            self.explore_tape = (torch.as_tensor(np.asarray(coins, np.float32), device=dev).contiguous(),
                                 torch.as_tensor(np.asarray(uniforms, np.float32), device=dev).contiguous())
            self._tape_pos = 0
        if indices is not None:
            self.index_tape = [np.asarray(i, np.int64) for i in indices]
            self._index_pos = 0

    def _tape_kw(self):
        if self.explore_tape is None:
            return {}
        k = self._tape_pos
        self._tape_pos += 1
        return dict(coin=self.explore_tape[0][k:k + 1], uniforms=self.explore_tape[1][k])

    def _train_epochs(self):
        """train_epochs (off_policy_marl.py:573-594): n_epochs x (sample, update)."""
        info = None
        if self.index_tape is not None:
            for _e in range(self.n_epochs):
                idx = self.index_tape[self._index_pos]
                self._index_pos += 1
                if idx.ndim == 2:
                    idx = idx[0] * self.memory.n_size + idx[1]
                info = self.learner.update(self.memory.sample(indexes=idx))
            return info
        if self.use_graph_updates:
            return self.learner.update_from_buffer(self.memory, self.n_epochs, seed=self.seed, sync=False)
        for _e in range(self.n_epochs):
            info = self.learner.update(self.memory.sample())
        return info

    def _train_rnn(self, train_steps):                         # off_policy_marl.py:335-349
        info, start = {}, self.current_step
        self.model._act_stale = True                            # (whatever happened to the parameters since the last call)
        while self.current_step - start < train_steps * self.n_envs:
            self.run_episodes(self.n_envs)
            if self.current_step >= self.start_training:
                info = self._train_epochs() or info
                self._cb("on_train_epochs_end", self.current_step, policy=self.model, memory=self.memory, train_steps=train_steps,
                         update_info=info)
        info = dict(self.learner.flush_info() or info)          # update phases ran unsynchronised: read the last one's info
        info["epsilon"] = self.e_greedy
        return info

    def train(self, train_steps):
        if self.use_rnn:
            return self._train_rnn(train_steps)
        env, n, N, A = self.envs, self.n_envs, self.n_agents, self.n_actions
        R = n * N
        if not self._started:
            env.reset()
            self._started = True
        info = {}
        two_buf = getattr(env, "double_buffered", False)       # the acted-on tensors survive step_device(): no copies
        fused_act = bool(getattr(self.config, "use_fused_acting", True))
        self.model._act_stale = True                            # (whatever happened to the parameters since the last call)
        for k in range(train_steps):
            if two_buf:
                obs, state, avail = env.buf_obs, env.buf_state, env.buf_avail
            else:
                obs, state, avail = env.buf_obs.clone(), env.buf_state.clone(), env.buf_avail.clone()
            # shared network on [n*N, obs] + the epsilon-greedy selection: one launch (xrl_marl_act_gru with H = 0, its weight
            # image kept current by the optimiser launch's mirrors), else three GEMM launches + xrl_marl_select_actions
            self._refresh_act_image(fused_act)
            if self.state_broadcast:                            # what the reference stores as this step's state (see __init__)
                # (k == 0: a train() call starts from the vector env's own buf_state, off_policy_marl.py:360)
                ops.marl_stored_state(state, env.done if k > 0 else None, self._stored_state)
                state = self._stored_state
            self.model.act_step(obs.view(R, -1), R, None, fused=fused_act and self.explore_tape is None,
                                select=dict(avail=avail if self.use_actions_mask else None, eps_dev=None, eps=float(self.e_greedy), action=env.action,
                                            action_f=self.act_f, seed=self.seed, step=self._host_step,
                                            step_dev=None, **self._tape_kw()))    # eager loop: the host knows the step index
            env.step_device()
            self._host_step += 1
            # off_policy_marl.py:363-371 (device tensors; the loop is a host loop already, so the hooks cost nothing unused)
            self._cb("on_train_step", self.current_step, envs=env, policy=self.model, obs=obs, actions=self.act_f, next_obs=env.next_obs,
                     rewards=env.rewards, terminals=env.terminals, agent_mask=env.agent_mask, state=state, next_state=env.next_state,
                     avail_actions=avail, next_avail_actions=env.next_avail, train_steps=train_steps)
            self.memory.store(obs=obs, actions=self.act_f, obs_next=env.next_obs, rewards=env.rewards,
                              terminals=env.terminals, agent_mask=env.agent_mask, state=state, state_next=env.next_state,
                              avail_actions=avail, avail_actions_next=env.next_avail)
            if self.current_step >= self.start_training and self.current_step % self.training_frequency == 0:
                info = self._train_epochs() or info
                self._cb("on_train_epochs_end", self.current_step, policy=self.model, memory=self.memory, train_steps=train_steps,
                         update_info=info)
            self.current_step += n
            self._update_explore_factor(push=False)
            self._cb("on_train_step_end", self.current_step, envs=env, policy=self.model, train_steps=train_steps, train_info=info)
        self._push_eps()
        info = dict(self.learner.flush_info() or info)          # update phases ran unsynchronised: read the last one's info
        info["epsilon"] = self.e_greedy
        return info

    # -- evaluation on HOST multi-agent vector envs (core/off_policy_marl.py:596-640 = run_episodes(test_mode=True), :426-560)
    @torch.no_grad()
    def greedy_actions(self, obs, avail=None, rnn=None):
        """obs [m, N, obs_dim] (NumPy / tensor), avail [m, N, A] or None -> int64 actions [m, N]: the masked argmax of the
        shared agent network (value_factorization.py:87-92), exploration off.  rnn: dict(h, c, reset) of device tensors
        [m*N, H] for recurrent agents (carried and reset row-wise by the recurrence kernel)."""
        dev, N, A = self.device, self.n_agents, self.n_actions
        X = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, device=dev).to(torch.float32)
        m = X.shape[0]
        R = m * N
        X = X.reshape(R, -1).contiguous()
        av = None
        if avail is not None and self.use_actions_mask:
            av = torch.as_tensor(np.asarray(avail) if not isinstance(avail, torch.Tensor) else avail, device=dev).to(torch.float32).reshape(R, A).contiguous()
        if self.use_rnn:
            self._refresh_act_image()
            q = self.model.act_step(X, R, rnn["h"], rnn["reset"], rnn.get("c"))
        else:
            q = self.model.agent_plan.forward(X, self.obs_dim, R)
        act = torch.zeros(R, dtype=torch.int32, device=dev)
        ops.marl_select_actions(q=q, avail=av, eps_dev=torch.zeros(1, device=dev), action=act, action_f=None, R=R, A=A, ld=A,
                                seed=self.seed, step=0, step_dev=None)
        return act.view(m, N).cpu().numpy().astype(np.int64)

    def test(self, test_episodes, test_envs=None, close_envs=True):
        """Episode scores (mean over agents) of `test_episodes` greedy episodes on `test_envs`: a vector env with the
        reference's multi-agent contract (dummy_vec_maenv.py:33-83: reset() -> (obs_list, infos); buf_avail_actions /
        buf_state; step(actions_list) -> (obs_list, rewards, terminated dicts, truncated, infos with reset_obs /
        reset_avail_actions / episode_score per agent))."""
        if test_envs is None:
            raise ValueError("`test_envs` must be provided for evaluation (the training envs live on the device).")
        keys, m = self.agent_keys, test_envs.num_envs
        obs_list, _ = test_envs.reset()
        avail = test_envs.buf_avail_actions if self.use_actions_mask else None
        rnn = None
        if self.use_rnn:
            z = lambda: torch.zeros(m * self.n_agents, self.model.RH, device=self.device)
            rnn = {"h": z(), "reset": torch.zeros(m * self.n_agents, device=self.device)}
            if self.model.lstm:
                rnn["c"] = z()
        scores, episodes = [], 0
        stack = lambda lst: np.stack([[np.asarray(d[k]) for k in keys] for d in lst])
        while episodes < test_episodes:
            acts = self.greedy_actions(stack(obs_list), stack(avail) if avail is not None else None, rnn)
            actions_list = [{k: int(acts[i, j]) for j, k in enumerate(keys)} for i in range(m)]
            next_obs, rewards, terminated, truncated, info = test_envs.step(actions_list)
            obs_list = list(next_obs)
            avail = list(test_envs.buf_avail_actions) if self.use_actions_mask else None
            ended = np.zeros(m, np.float32)
            for i in range(m):
                if all(terminated[i].values()) or truncated[i]:
                    episodes += 1
                    obs_list[i] = info[i]["reset_obs"]
                    if avail is not None:
                        avail[i] = info[i]["reset_avail_actions"]
                    ended[i] = 1.0                                         # init_rnn_states_item (:504-505)
                    scores.append(float(np.mean([info[i]["episode_score"][k] for k in keys])))
            if rnn is not None:
                self._set_reset_rows(rnn["reset"], torch.from_numpy(ended).to(self.device), m, self.n_agents)
        self.log_infos({"Test-Results/Episode-Rewards": float(np.mean(scores)),
                        "Test-Results/Episode-Rewards-Std": float(np.std(scores))}, self.current_step)
        if close_envs:
            test_envs.close()
        return scores


class VDN_Agents(QMIX_Agents):
    """xuance/torch/agents/multi_agent_rl/vdn_agents.py: the QMIX loop with VDN_Mixer (sum) and VDN_Learner."""
    mixer_name, learner_cls = "VDN", VDN_Learner


class IQL_Agents(QMIX_Agents):
    """xuance/torch/agents/multi_agent_rl/iql_agents.py: independent Q-learners (IndependentMixer, IQL_Learner)."""
    mixer_name, learner_cls = "Independent", IQL_Learner
    eps_decay_per_env = False

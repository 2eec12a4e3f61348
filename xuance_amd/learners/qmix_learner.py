"""QMIX learner (feed-forward agents, parameter sharing) on the HIP engine.

Same constructor ``(config, agent_grouping_or_keys, model, callback)``, ``update(sample)`` contract, info keys
(``learning_rate, loss_Q, predictQ``) and callback hooks as
xuance/torch/learners/multi_agent_rl/qmix_learner.py:13-112 (with iql_learner.py:37-83 and
base/marl_learner.py:319-408).  ``sample`` may be the reference buffer's nested dict (field -> agent -> array
[B, ...]) or the stacked form produced by HipMARLOffPolicyBuffer (field -> tensor [B, N, ...]).

Recurrent agents (use_rnn, the default of configs/qmix/sc2/3m.yaml): ``sample`` is what
MARL_OffPolicyBuffer_RNN.sample returns (memory_tools_marl.py:970-996: obs [B, T+1, ...] per agent, actions / rewards /
terminals / agent_mask [B, T], avail_actions [B, T+1, A], state [B, T+1, S], filled [B, T]).  On the device the batch is
TIME-MAJOR (row t*B*N + b*N + n): every layer outside the recurrence is one GEMM over all (T+1)*B*N rows, "next step"
tensors are the same tensors one slot further, and the mixer / TD kernel treats the T*B (step, episode) pairs as its
batch.  Two properties of the reference's recurrent branch (both pinned by fixtures, oracle/make_golden.py):
  * iql_learner.py:58 re-slices q_eval inside torch.no_grad(): the agent networks get no gradient, only the mixer
    trains.  ``config.rnn_backprop_agents`` (default False) keeps that behaviour; True back-propagates through the Q
    head, the GRU (xrl_gru_backward) and the fc layer.
  * iql_learner.py:78-81 raises IndexError when use_actions_mask is on (agent axis sliced instead of time); here the
    mask of step t+1 is applied to step t's target, as the feed-forward branch does.
"""
import numpy as np
import torch

from .. import ops
from .base import _NullCallback, Learner, AdamHandle, LinearLRHandle
from .ppo_learner import pick_n_split


class QMIX_Learner(Learner):
    mixer_mode = 0          # xrl_qmix_t.mixer: 0 QMIX, 1 VDN (sum), 2 independent (IQL)

    def __init__(self, config, agent_grouping, model, callback=None):
        keys = list(getattr(agent_grouping, "agent_keys", agent_grouping))
        super().__init__(config, model, callback,
                         adopt_hints=dict(n_agents=len(keys), mixer=("QMIX", "VDN", "Independent")[self.mixer_mode]))
        model = self.model                                          # (a reference nn.Module was adopted by the base class)
        assert {"QMIX": 0, "VDN": 1, "Independent": 2}[getattr(model, "mixer", "QMIX")] == self.mixer_mode, \
            "the model's mixer does not match the learner"
        if self.mixer_mode and getattr(config, "use_rnn", False):
            # the reference's recurrent branch detaches q_eval (iql_learner.py:49,58): without a trainable mixer nothing
            # requires grad there and its loss.backward() raises
            raise NotImplementedError("VDN / IQL with recurrent agents: the reference's own update raises in this configuration")
        self.use_rnn = bool(getattr(config, "use_rnn", False))
        assert self.use_rnn == bool(getattr(model, "use_rnn", False)), "config.use_rnn and the model disagree"
        self.rnn_backprop_agents = bool(getattr(config, "rnn_backprop_agents", False))
        self.agent_keys = list(getattr(agent_grouping, "agent_keys", agent_grouping))
        self.n_agents = len(self.agent_keys)
        assert self.n_agents == model.n_agents
        self.use_parameter_sharing = getattr(config, "use_parameter_sharing", True)
        self.sync_frequency = config.sync_frequency
        self.double_q = bool(getattr(config, "double_q", True))
        P = model.params
        self.optimizer = AdamHandle(P, model.trainable_order, self.learning_rate, eps=1e-5,
                                    weight_decay=getattr(config, "weight_decay", 0.0), total_iters=self.total_iters,
                                    end_factor=self.end_factor_lr_decay)
        self.scheduler = LinearLRHandle(self.optimizer)
        dev = P.device
        self._cap = 0
        self.sumsq = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.sums = torch.zeros(8, dtype=torch.float64, device=dev)
        self.opt_sync = torch.zeros(4 + (P.P + 255) // 256 + 8, dtype=torch.int32, device=dev)   # barrier scratch of xrl_reduce_adam
        self.sync_replicas_from_rank0()

    def estimate_total_iterations(self):                        # marl_learner.py:36-47 (feed-forward branch)
        c = self.config
        start_training = getattr(c, "start_training", 0)
        training_frequency = getattr(c, "training_frequency", 1)
        if getattr(c, "use_rnn", False):                           # marl_learner.py:42-43
            return (c.running_steps - start_training) // (self.episode_length * c.parallels) * getattr(c, "n_epochs", 1)
        return (c.running_steps - start_training) // (training_frequency * c.parallels) * getattr(c, "n_epochs", 1)

    def _ensure(self, B):
        if B <= self._cap:
            return
        m, dev = self.model, self.model.params.device
        self._cap = B
        N, R = m.n_agents, B * m.n_agents
        self.slabs = torch.zeros(32, m.params.P, device=dev)
        self.partials = torch.zeros(B, 8, dtype=torch.float64, device=dev)
        self.diag = torch.zeros(3 * B * N, device=dev)
        self.X = torch.zeros(2 * R, m.obs_dim, device=dev)       # rows [0,R) obs, rows [R,2R) obs_next
        self.states = torch.zeros(2 * B, m.state_dim, device=dev)
        self.buf = {k: torch.zeros(B, N, device=dev) for k in ("actions", "rewards", "terminals", "agent_mask")}
        self.buf["avail_next"] = torch.ones(B, N, m.n_actions, device=dev)
        m.agent_plan.ensure(2 * R)
        m.agent_target_plan.ensure(R)
        if self.mixer_mode == 0:
            m.mixer_plan.ensure(B)
            m.mixer_target_plan.ensure(B)

    def _stack(self, x, dtype=torch.float32):
        """field -> agent -> [B, ...]  (reference buffers)  or an already stacked [B, N, ...] tensor/array."""
        dev = self.model.params.device
        if isinstance(x, dict):
            return torch.stack([torch.as_tensor(np.asarray(x[k]) if not isinstance(x[k], torch.Tensor) else x[k],
                                                device=dev).to(dtype) for k in self.agent_keys], dim=1)
        return torch.as_tensor(x, device=dev).to(dtype)

    def build_training_data(self, sample):                      # marl_learner.py:319-408
        B = int(sample["batch_size"])
        self._ensure(B)
        m = self.model
        R = B * m.n_agents
        self.X[:R].copy_(self._stack(sample["obs"]).reshape(R, -1))
        self.X[R:2 * R].copy_(self._stack(sample["obs_next"]).reshape(R, -1))
        for k in ("actions", "rewards", "terminals", "agent_mask"):
            self.buf[k][:B].copy_(self._stack(sample[k]).reshape(B, m.n_agents))
        if self.use_actions_mask:
            self.buf["avail_next"][:B].copy_(self._stack(sample["avail_actions_next"]).reshape(B, m.n_agents, -1))
        dev = m.params.device
        if self.mixer_mode == 0:                               # only the QMIX mixer reads the global state
            self.states[:B].copy_(torch.as_tensor(sample["state"], device=dev).reshape(B, -1))
            self.states[B:2 * B].copy_(torch.as_tensor(sample["state_next"], device=dev).reshape(B, -1))
        return B

    _images_current = False      # True only inside an update phase (the images are known to be current there)
    _images_stale = True         # something other than the optimiser launch's mirrors may have changed the parameters
    _images_version = -1         # the model's `version` the images were last rebuilt at

    def fused_eligible(self):
        """The one-launch update (xrl_qmix_fused_update): QMIX mixer, feed-forward agents of at most 4 linear layers with one
        hidden activation, mixer embed dim <= 64, activations of a group of transitions within the 160 KB of LDS."""
        if not hasattr(self, "_fused"):
            self._fused = None
            m = self.model
            ok = self.mixer_mode == 0 and not m.use_rnn and getattr(self.config, "use_fused_qmix_update", True) and \
                1 <= len(m.agent_plan.stages) <= 4 and all(len(s) == 1 for s in m.agent_plan.stages) and \
                len({s[0].act for s in m.agent_plan.stages[:-1]}) <= 1 and m.H <= 64 and m.n_agents <= 64
            if ok:
                # transitions per workgroup: config.fused_qmix_items_per_wg, default ONE (32 workgroups, VALU products).  Larger groups
                # run their products as matrix-core tiles (5 x 3 agents = 15 rows of a 16-row tile: 7 workgroups / slabs per update) --
                # measured round 4 (profiles/r04_b_qmix_ff_mfma.txt): parity identical, the launch 25 -> 34 us: the update's 9 MFLOP on
                # 7 CUs are bound by the per-CU fp32 MFMA rate (~2.5 k cycles per product phase) where 32 CUs share them at 3 rows each.
                want = getattr(self.config, "fused_qmix_items_per_wg", None)
                for items in ([int(want)] if want else [1]):
                    # products: 2 = VALU (the default, whatever n_agents is: the matrix-core form measured slower, and its automatic
                    # choice -- 0 -- would engage for any team of >= 8 agents at one transition per workgroup); 1 / 0 opt in
                    fs = ops.QmixFusedState(m, self.double_q, self.gamma, items, int(getattr(self.config, "fused_qmix_products", 2)))
                    if 0 < fs.lds_bytes() <= 160 * 1024:
                        self._fused = fs
                        break
        return self._fused is not None

    def _step(self, B, ring=None):
        """ring: None, or (memory, draw arguments): the one-launch update then reads its batch from the replay ring itself."""
        m, opt = self.model, self.optimizer
        N, A, H = m.n_agents, m.n_actions, m.H
        R = B * N
        if self.fused_eligible() and self._fused.n_groups(B) <= self.slabs.shape[0]:
            b = self.buf
            if not self._images_current:                    # (a call outside a captured phase: somebody may have written the
                self._fused.refresh()                       #  parameters -- load_model, copy_target, an adopter's module)
            if ring is not None:
                f = ring["memory"].soa.fields
                S = ops.qmix_fused_update(self._fused, B, f["obs"], f["obs_next"], f["state"], f["state_next"], f["actions"],
                                          f["rewards"], f["terminals"], f["agent_mask"],
                                          f["avail_actions_next"] if self.use_actions_mask else None, self.slabs, m.params.P,
                                          self.partials, self.diag, ring=ring)
            else:
                S = ops.qmix_fused_update(self._fused, B, self.X, self.X[R:], self.states, self.states[B:], b["actions"],
                                          b["rewards"], b["terminals"], b["agent_mask"],
                                          b["avail_next"] if self.use_actions_mask else None, self.slabs, m.params.P,
                                          self.partials, self.diag)
            self._finish_step(S)
            return
        S = pick_n_split(R)
        from ..nets import Plan
        # eval network on obs (+ next_obs for the double-Q argmax) and target network on next_obs (iql_learner.py:41-47,
        # 63-71); eval hyper-networks on state, target ones on state_next.  The four plans are independent until the
        # mixer needs them all, so layer i of each goes into ONE grouped launch (3 launches for the whole forward)
        items = [(m.agent_plan, self.X, m.obs_dim, 2 * R if self.double_q else R, None),
                 (m.agent_target_plan, self.X[R:], m.obs_dim, R, m.target_flat)]
        if self.mixer_mode == 0:
            items += [(m.mixer_plan, self.states, m.state_dim, B, None),
                      (m.mixer_target_plan, self.states[B:], m.state_dim, B, m.target_flat)]
        outs = Plan.forward_many(items)
        q_all, q_next = outs[0], outs[1]
        d_q = m.agent_plan.dacts[len(m.agent_plan.widths) - 1]
        common = dict(q_eval=q_all, q_next_eval=q_all[R:] if self.double_q else None, q_next=q_next,
                      actions=self.buf["actions"], avail_next=self.buf["avail_next"] if self.use_actions_mask else None,
                      agent_mask=self.buf["agent_mask"], rewards=self.buf["rewards"], terminals=self.buf["terminals"],
                      d_q=d_q, diag=self.diag, partials=self.partials, B=B, N=N, A=A, ldq=A, double_q=int(self.double_q),
                      gamma=float(self.gamma), mixer=self.mixer_mode)
        back = [(m.agent_plan, self.X, m.obs_dim, R)]
        if self.mixer_mode == 0:
            e_raw, t_raw = outs[2], outs[3]
            e_l1, t_l1 = m.mixer_plan.acts[1], m.mixer_target_plan.acts[1]
            d_l1, d_raw = m.mixer_plan.dacts[1], m.mixer_plan.dacts[2]
            ld1, ld2 = m.mixer_plan.widths[1], m.mixer_plan.widths[2]
            ops.qmix_mix_td(e_b1=e_l1.data_ptr() + 4 * 3 * m.HH, e_raw=e_raw, t_b1=t_l1.data_ptr() + 4 * 3 * m.HH, t_raw=t_raw,
                            d_e_b1=d_l1.data_ptr() + 4 * 3 * m.HH, d_e_raw=d_raw, H=H, ld_e1=ld1, ld_e2=ld2, ld_t1=ld1,
                            ld_t2=ld2, **common)
            back.append((m.mixer_plan, self.states, m.state_dim, B))
        else:
            ops.qmix_mix_td(**common)                       # VDN: sum mixer; IQL: per-agent TD (no hyper-networks)
        # data-gradient chains of both plans side by side, then every weight gradient of the update as one grouped launch
        Plan.backward_many(back, self.slabs, S)
        self._finish_step(S)

    def _finish_step(self, S):
        """Slab reduction, gradient norm, clip, Adam, LinearLR and the periodic hard target update (qmix_learner.py:88-106):
        ONE launch (xrl_reduce_adam with the target as a periodic mirror) on a single GPU; with several ranks the flat
        gradient is all-reduced between the reduction and the optimiser launch."""
        m, opt, P = self.model, self.optimizer, self.model.params.P
        clip = self.grad_clip_norm if self.use_grad_clip else 0.0
        fs = getattr(self, "_fused", None)
        if not self.needs_collective() and self._fused_optimizer_ok(self.gradient_exchange() is not None):
            # (the one-launch update reads weight images: the optimiser launch writes every new parameter there as well)
            mirrors = [(fs.map, fs.img_eval)] if fs else []
            act = getattr(m, "_act_state", None)            # the acting launch's weight image follows every step too
            if act is not None:
                mirrors.append((act.map, act.image))
            tail = getattr(self, "_tail", None) or {}           # (update_from_buffer: this epoch's loss sums, the phase's draw-counter tick)
            ops.reduce_adam(self.slabs, S, P, m.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip,
                            mirrors, self.opt_sync, target=m.target_flat,
                            target_every=self.sync_frequency, exchange=self.gradient_exchange(),
                            target_image=fs.img_target if fs else None, **tail)
            if tail:
                self._tail_done += 1
            return
        self._images_current = False
        self._images_stale = True
        if getattr(m, "_act_state", None) is not None:
            m._act_stale = True                             # (this path has no mirrors: the agents refresh before acting)
        ops.grad_reduce(self.slabs, S, P, P, opt.grad, self.sumsq)
        if self.distributed_training and self.world_size > 1:
            from ..dist import allreduce_mean_
            allreduce_mean_(opt.grad)
            ops.grad_reduce(opt.grad, 1, P, P, opt.grad, self.sumsq)
        ops.adam_step(m.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip)
        ops.sync_target(m.params.flat, m.target_flat, P, opt.state, self.sync_frequency)

    # ------------------------------------------------------------------ recurrent branch
    def _ensure_rnn(self, B, T):
        if getattr(self, "_cap_rnn", None) == (B, T):
            return
        m, dev = self.model, self.model.params.device
        self._cap_rnn = (B, T)
        N, T1, R = m.n_agents, T + 1, B * m.n_agents
        self.slabs = torch.zeros(32, m.params.P, device=dev)
        self.partials = torch.zeros(T * B, 8, dtype=torch.float64, device=dev)
        self.diag = torch.zeros(3 * T * B, device=dev)
        self.Xs = torch.zeros(T1 * R, m.obs_dim, device=dev)                  # [t][b][n][obs]
        self.states_s = torch.zeros(T1 * B, m.state_dim, device=dev)          # [t][b][state]
        self.seq = {k: torch.zeros(T, B, N, device=dev) for k in ("actions", "rewards", "terminals", "agent_mask")}
        self.seq["avail"] = torch.ones(T1, B, N, m.n_actions, device=dev)
        self.seq["filled"] = torch.zeros(T, B, device=dev)
        m.seq_workspace(0, R, T1)
        m.seq_workspace(1, R, T1)
        m.post_plans[0].dacts[len(m.post_plans[0].widths) - 1].zero_()
        m.mixer_plan.ensure(T1 * B)
        m.mixer_target_plan.ensure(T1 * B)

    def build_training_data_rnn(self, sample):                  # marl_learner.py:319-408, recurrent branch
        B, T = int(sample["batch_size"]), int(sample["sequence_length"])
        self._ensure_rnn(B, T)
        m, dev = self.model, self.model.params.device
        N, T1 = m.n_agents, T + 1
        self.Xs.view(T1, B, N, -1).copy_(self._stack(sample["obs"]).reshape(B, N, T1, -1).permute(2, 0, 1, 3))
        for k in ("actions", "rewards", "terminals", "agent_mask"):
            self.seq[k].copy_(self._stack(sample[k]).reshape(B, N, T).permute(2, 0, 1))
        if self.use_actions_mask:
            self.seq["avail"].copy_(self._stack(sample["avail_actions"]).reshape(B, N, T1, -1).permute(2, 0, 1, 3))
        self.states_s.view(T1, B, -1).copy_(torch.as_tensor(sample["state"], device=dev).to(torch.float32)
                                            .reshape(B, T1, -1).permute(1, 0, 2))
        self.seq["filled"].copy_(torch.as_tensor(sample["filled"], device=dev).to(torch.float32).reshape(B, T).t())
        return B, T

    def _step_rnn(self, B, T):
        m, opt = self.model, self.optimizer
        N, A, H, T1 = m.n_agents, m.n_actions, m.H, T + 1
        R, BT = B * N, T * B
        S = pick_n_split(T1 * R)
        from ..nets import Plan
        # agent networks over the sequences (iql_learner.py:39-47, 53-57); the hyper-networks of the eval mixer (slots
        # 0..T-1 are used) and of the target mixer (slots 1..T) ride in the launches of the layers above the recurrence
        q_all, q_tgt, e_raw, t_raw = m.agent_forward_seq_pair(
            self.Xs, R, T1, ride_along=[(m.mixer_plan, self.states_s, m.state_dim, T1 * B, None),
                                        (m.mixer_target_plan, self.states_s, m.state_dim, T1 * B, m.target_flat)])
        e_l1, t_l1 = m.mixer_plan.acts[1], m.mixer_target_plan.acts[1]
        d_l1, d_raw = m.mixer_plan.dacts[1], m.mixer_plan.dacts[2]
        ld1, ld2 = m.mixer_plan.widths[1], m.mixer_plan.widths[2]
        post = m.post_plans[0]
        d_q = post.dacts[len(post.widths) - 1]
        # (rows of the last slot of d_q stay zero: Q of slot T is not in the loss (:58) and nothing ever writes them)
        ops.qmix_mix_td(q_eval=q_all, q_next_eval=q_all[R:] if self.double_q else None, q_next=q_tgt[R:],
                        actions=self.seq["actions"], avail_next=self.seq["avail"][1:] if self.use_actions_mask else None,
                        agent_mask=self.seq["agent_mask"], rewards=self.seq["rewards"], terminals=self.seq["terminals"],
                        e_b1=e_l1.data_ptr() + 4 * 3 * m.HH, e_raw=e_raw, t_b1=t_l1[B:].data_ptr() + 4 * 3 * m.HH, t_raw=t_raw[B:],
                        d_q=d_q, d_e_b1=d_l1.data_ptr() + 4 * 3 * m.HH, d_e_raw=d_raw, diag=self.diag, partials=self.partials,
                        B=BT, N=N, A=A, H=H, ldq=A, ld_e1=ld1, ld_e2=ld2, ld_t1=ld1, ld_t2=ld2, double_q=int(self.double_q),
                        gamma=float(self.gamma), filled=self.seq["filled"])
        wg = []
        m.mixer_plan.backward(self.states_s, m.state_dim, BT, self.slabs, S, defer_wgrad=wg)
        ops.linear_bwd_weight(wg, S, self.slabs.shape[1])
        if self.rnn_backprop_agents:
            m.agent_backward_seq(self.Xs, R, T1, self.slabs, S)
        self._finish_step(S)

    def _info_rnn(self, B, T, sums):
        """loss = sum((td * filled)^2) / sum(filled) (qmix_learner.py:82-84); predictQ = mean over all B*T (:101)."""
        return {"learning_rate": self.read_optimizer().last_lr, "loss_Q": float(sums[0] / sums[2]),
                "predictQ": float(sums[1] / (B * T))}

    def _cb_rnn(self, B, T):
        d = self.diag.view(3, T, B).transpose(1, 2).reshape(3, B * T)          # the reference's (episode, step) order
        return dict(q_tot_eval=d[0], q_tot_next=d[1], q_tot_target=d[2])

    # ------------------------------------------------------------------ whole update phases straight from the HBM replay buffer
    def update_from_buffer(self, memory, n_epochs=1, seed=1, sync=True):
        """`n_epochs` updates (sample -> gather -> forward / mixer TD / backward -> Adam -> target sync) as ONE captured
        hipGraph launch: indices are drawn on the device (xrl_sample_replay_indices, following the filling ring through
        memory.size_dev), the gather writes straight into the staging tensors the networks read, and the loss terms of
        every update are read back with a single host sync at the end.  Same arithmetic as update(memory.sample()).
        With `sync=False` (and no user callback to serve) the call returns None right after the launch; `flush_info()`
        later returns the info of the last phase launched."""
        if self.use_rnn:
            return self._update_from_episodes(memory, n_epochs, seed, sync)
        B, m, dev = memory.batch_size, self.model, self.model.params.device
        key = (id(memory), n_epochs, B)
        ver = getattr(m, "version", 0)
        if self.fused_eligible() and (self._images_stale or ver != self._images_version):   # (eager, outside the captured
            self._fused.refresh()                           # phase: load_state_dict / copy_target bump the model's version,
            self._images_stale, self._images_version = False, ver   # load_model and the unfused optimiser path set the flag)
        if getattr(self, "_buf_graph_key", None) != key:
            self._ensure(B)
            R, N = B * m.n_agents, m.n_agents
            self._idx = torch.zeros(B, dtype=torch.int64, device=dev)
            if getattr(self, "_sample_counter", None) is None:      # (kept across re-captures: the replay draws' Philox counter)
                self._sample_counter = torch.zeros(1, dtype=torch.int32, device=dev)
            self._epoch_sums = torch.zeros(n_epochs, 8, dtype=torch.float64, device=dev)
            dst = {"obs": self.X[:R].view(B, -1), "obs_next": self.X[R:2 * R].view(B, -1), "actions": self.buf["actions"][:B],
                   "rewards": self.buf["rewards"][:B], "terminals": self.buf["terminals"][:B],
                   "agent_mask": self.buf["agent_mask"][:B], "state": self.states[:B], "state_next": self.states[B:2 * B]}
            if self.use_actions_mask:
                dst["avail_actions_next"] = self.buf["avail_next"][:B].view(B, -1)

            self._phase_partials = torch.zeros(n_epochs, B, 8, dtype=torch.float64, device=dev)
            self._phase_sumsq = torch.zeros(64, dtype=torch.float64, device=dev)
            self._phase_scalars = torch.zeros(2 * max(n_epochs, 1), device=dev)
            if getattr(self, "_phase_sync", None) is None:
                self._phase_sync = torch.zeros(ops._lib.QF_PHASE_SYNC_WORDS, dtype=torch.int32, device=dev)

            def phase_launch_ok():
                """The whole phase as ONE launch (xrl_qmix_fused_phase, round 6): the one-launch update reading its batches from the ring,
                no gradient clipping, one rank, the one-launch optimiser usable, the workgroups resident on one XCD.  OFF unless
                config.use_qmix_phase_launch: True -- bit-identical to the launch pairs (tests/test_gpu_offpolicy_agents.py) and measured
                slower: 42.9 us per update against 34.8 us (profiles/r06_d_qmix_phase.json; DESIGN.md section 3 "Round 6": 32 workgroups
                on ONE XCD pull slabs / parameters at ~10 B/clk per CU, the separate optimiser launch spreads them over 68 CUs and 8 L2s)."""
                fs = self._fused if self.fused_eligible() else None
                return fs is not None and bool(getattr(self.config, "use_qmix_phase_launch", False)) and not self.use_grad_clip and \
                    getattr(self.config, "fused_qmix_gather_in_kernel", True) and fs.n_groups(B) <= self.slabs.shape[0] and \
                    int(fs.struct.products) == 2 and not self.needs_collective() and self.gradient_exchange() is None and \
                    self._fused_optimizer_ok(False) and ops.qmix_fused_phase_fits(B, fs.items_per_wg, m.params.P)

            def enqueue_phase():
                fs, opt, P = self._fused, self.optimizer, m.params
                self._images_current = True
                act = getattr(m, "_act_state", None)            # the acting launch's weight image follows every step too
                f = memory.soa.fields
                ops.qmix_fused_phase(fs, B, f, f["avail_actions_next"] if self.use_actions_mask else None, self.slabs, P.P, self.diag,
                                     dict(n_envs=memory.n_envs, n_size=memory.n_size, size_dev=memory.size_dev, seed=seed, counter=0,
                                          counter_dev=self._sample_counter, idx_out=self._idx),
                                     dict(n_updates=n_epochs, sync_every=self.sync_frequency, params=P.flat, grad=opt.grad, m=opt.m, v=opt.v, P=P.P,
                                          state=opt.state, map=fs.map, target=m.target_flat, act_image=act.image if act is not None else None,
                                          act_map=act.map if act is not None else None, phase_partials=self._phase_partials,
                                          epoch_sums=self._epoch_sums, sumsq_part=self._phase_sumsq, scalars=self._phase_scalars,
                                          tick=self._sample_counter, tick_inc=n_epochs, sync=self._phase_sync))
                self.partials = self._phase_partials[n_epochs - 1]
                self._images_current = False

            def enqueue():
                # per update: draw, gather, step; the draw counter and the loss sums are settled once per phase
                if self._phase_launch:
                    return enqueue_phase()
                if self.fused_eligible():                   # weight images of the one-launch update: kept current by the
                    self._images_current = True             # optimiser launch's mirrors (update_from_buffer refreshed them
                                                            # before this phase if anything else touched the parameters)
                in_kernel = self.fused_eligible() and self._fused.n_groups(B) <= self.slabs.shape[0] and \
                    getattr(self.config, "fused_qmix_gather_in_kernel", True)
                self._tail_done = 0
                for e in range(n_epochs):
                    self.partials = self._phase_partials[e]
                    # this epoch's loss sums -- and, with the last epoch, the phase's draw-counter tick -- ride in the optimiser
                    # launch (xrl_mirrors_t.part / .tick) instead of two launches behind the phase
                    self._tail = dict(partials=(self._phase_partials[e], B, self._epoch_sums[e]))
                    if e == n_epochs - 1:
                        self._tail["tick"] = (self._sample_counter, n_epochs)
                    try:
                        if in_kernel:                       # the update launch draws and gathers its own batch from the ring
                            self._step(B, ring=dict(memory=memory, n_envs=memory.n_envs, n_size=memory.n_size,
                                                    size_dev=memory.size_dev, seed=seed, counter=e,
                                                    counter_dev=self._sample_counter, idx_out=self._idx))
                        else:
                            memory.draw_into(self._idx, dst, seed, e, self._sample_counter)     # draw + gather: one launch
                            self._step(B)
                    finally:
                        self._tail = None
                self._images_current = False
                if self._tail_done != n_epochs:             # (the two-launch optimiser path: nothing rode along)
                    assert self._tail_done == 0
                    ops.counter_add(self._sample_counter, n_epochs)
                    ops.sum_partials_batched(self._phase_partials, B, 8, self._epoch_sums, n_epochs, B * 8, 8)
            self._phase_launch = phase_launch_ok() and n_epochs <= 64
            self._buf_enqueue, self._buf_graph, self._buf_graph_key = enqueue, None, key
            enqueue()                                       # this call's phase runs eagerly (lazy allocations happen here) ...
            if not self.needs_collective():
                torch.cuda.synchronize()                    # ... and is then captured for the following calls (the gradient
                g = ops.Graph()                             #     all-reduce of the multi-GPU path cannot be captured)
                with g:
                    enqueue()
                self._buf_graph = g
        elif self._buf_graph is not None:
            self._buf_graph.launch()
        else:
            self._buf_enqueue()
        return self._finish_phase(("ff", n_epochs, B, None), sync)

    def _finish_phase(self, phase, sync):
        self._pending_phase = phase
        if not sync and isinstance(self.callback, _NullCallback):
            self.iterations += phase[1]
            return None
        return self._phase_info(count=True)

    def flush_info(self):
        """Info of the last update phase launched with sync=False ({} if there was none)."""
        return self._phase_info(count=False) if getattr(self, "_pending_phase", None) else {}

    def _phase_info(self, count):
        kind, n_epochs, B, T = self._pending_phase
        self._pending_phase = None
        sums = self._epoch_sums.cpu().numpy()               # the one host sync of the phase
        st = self.read_optimizer() if kind == "ff" else None
        info = {}
        for e in range(n_epochs):
            self.iterations += int(count)
            info = self.callback.on_update_start(self.iterations, model=self.policy) or {}
            if kind == "ff":
                info.update(self._info_ff(B, sums[e], st.last_lr))
                cb = self._cb_ff(B)
            else:
                info.update(self._info_rnn(B, T, sums[e]))
                cb = self._cb_rnn(B, T)
            info.update(self.callback.on_update_end(self.iterations, model=self.policy, info=info, **cb) or {})
        return info

    def _info_ff(self, B, sums, lr):                            # qmix_learner.py:98-102
        return {"learning_rate": lr, "loss_Q": float(sums[0] / B), "predictQ": float(sums[1] / B)}

    def _cb_ff(self, B):                                        # :108-110
        return dict(q_tot_eval=self.diag[:B], q_tot_next=self.diag[B:2 * B], q_tot_target=self.diag[2 * B:3 * B])

    def _update_from_episodes(self, memory, n_epochs, seed, sync=True):
        """Recurrent twin of update_from_buffer: episodes are drawn on the device (uniform over memory.size_dev, as
        np.random.choice(size, batch_size) does, memory_tools_marl.py:981), gathered time-major straight into the staging
        tensors (xrl_episode_gather) and the whole n_epochs phase replays as one hipGraph."""
        B, T, m, dev = memory.batch_size, memory.max_eps_len, self.model, self.model.params.device
        key = (id(memory), n_epochs, B, T)
        if getattr(self, "_buf_graph_key", None) != key:
            self._ensure_rnn(B, T)
            T1, N = T + 1, m.n_agents
            self._idx = torch.zeros(B, dtype=torch.int64, device=dev)
            if getattr(self, "_sample_counter", None) is None:      # (kept across re-captures: the replay draws' Philox counter)
                self._sample_counter = torch.zeros(1, dtype=torch.int32, device=dev)
            self._epoch_sums = torch.zeros(n_epochs, 8, dtype=torch.float64, device=dev)
            dst = {"obs": self.Xs.view(T1, B, -1), "actions": self.seq["actions"], "rewards": self.seq["rewards"],
                   "terminals": self.seq["terminals"], "agent_mask": self.seq["agent_mask"],
                   "state": self.states_s.view(T1, B, -1), "filled": self.seq["filled"].view(T, B, 1)}
            if self.use_actions_mask:
                dst["avail_actions"] = self.seq["avail"].view(T1, B, -1)

            self._phase_partials = torch.zeros(n_epochs, T * B, 8, dtype=torch.float64, device=dev)

            def enqueue():
                for e in range(n_epochs):
                    memory.draw_into(self._idx, dst, seed, e, self._sample_counter)      # draw + gather: one launch
                    self.partials = self._phase_partials[e]
                    self._step_rnn(B, T)
                ops.counter_add(self._sample_counter, n_epochs)
                ops.sum_partials_batched(self._phase_partials, T * B, 8, self._epoch_sums, n_epochs, T * B * 8, 8)
            self._buf_enqueue, self._buf_graph, self._buf_graph_key = enqueue, None, key
            enqueue()
            if not self.needs_collective() and getattr(self.config, "use_hip_graph", True):
                torch.cuda.synchronize()
                g = ops.Graph()
                with g:
                    enqueue()
                self._buf_graph = g
        elif self._buf_graph is not None:
            self._buf_graph.launch()
        else:
            self._buf_enqueue()
        return self._finish_phase(("rnn", n_epochs, B, T), sync)

    def update(self, sample):
        self.iterations += 1
        if self.use_rnn:
            B, T = self.build_training_data_rnn(sample)
            info = self.callback.on_update_start(self.iterations, model=self.policy) or {}
            self._step_rnn(B, T)
            ops.sum_partials(self.partials, T * B, 8, self.sums)
            info.update(self._info_rnn(B, T, self.sums.cpu().numpy()))
            info.update(self.callback.on_update_end(self.iterations, model=self.policy, info=info, **self._cb_rnn(B, T)) or {})
            return info
        B = self.build_training_data(sample)
        info = self.callback.on_update_start(self.iterations, model=self.policy) or {}
        self._step(B)
        ops.sum_partials(self.partials, B, 8, self.sums)
        info.update(self._info_ff(B, self.sums.cpu().numpy(), self.read_optimizer().last_lr))
        info.update(self.callback.on_update_end(self.iterations, model=self.policy, info=info, **self._cb_ff(B)) or {})
        return info


class VDN_Learner(QMIX_Learner):
    """Value decomposition networks (xuance/torch/learners/multi_agent_rl/vdn_learner.py:13-106): QMIX_Learner with
    VDN_Mixer, the parameter-free sum over the (masked) agent values -- `xrl_qmix_t.mixer = 1`, no hyper-network launches.
    The model is MixingQNet(mixer="VDN") (vdn_agents.py:71-79)."""
    mixer_mode = 1


class IQL_Learner(QMIX_Learner):
    """Independent Q-learning (iql_learner.py:13-142): per-agent TD error on the unmixed values, loss
    sum((td * mask)^2) / sum(mask), info keys prefixed with the group name -- `xrl_qmix_t.mixer = 2`.  With parameter
    sharing there is one group, hence one optimiser (iql_learner.py:24-36 builds one per group).
    The model is MixingQNet(mixer="Independent") (iql_agents.py:71-79)."""
    mixer_mode = 2

    def _info_ff(self, B, sums, lr):                            # iql_learner.py:129-135
        g = self.model.group
        return {f"{g}/learning_rate": lr, f"{g}/loss_Q": float(sums[0] / sums[2]),
                f"{g}/predictQ": float(sums[1] / (B * self.n_agents))}

    def _cb_ff(self, B):                                        # :140 passes no tensors
        return {}

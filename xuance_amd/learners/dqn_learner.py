"""DQN learner on the HIP engine: same constructor / ``update(**samples)`` contract / info keys / callback hooks as
xuance/torch/learners/qlearning_family/dqn_learner.py:12-75 (MSE TD loss -- or, with ``use_huber_loss``, nn.HuberLoss(
``huber_delta``) in its place --, max-target, hard target sync).  With
``double_q=True`` the target action comes from the eval network (the DDQN rule, ddqn_learner.py:39-47)."""
import torch

from .. import ops
from .base import _NullCallback, Learner, AdamHandle, LinearLRHandle
from .ppo_learner import pick_n_split


class DQN_Learner(Learner):
    def __init__(self, config, model, callback=None):
        super().__init__(config, model, callback)
        model = self.model                                          # (a reference nn.Module was adopted by the base class)
        self.sync_frequency = config.sync_frequency
        self.double_q = bool(getattr(config, "double_q", False))
        # TD loss: nn.MSELoss as in the reference's DQN family (dqn_learner.py:25,46) unless config.use_huber_loss asks for
        # nn.HuberLoss(delta = config.huber_delta) -- the switch and its names as the reference's learners carry them
        # (learners/base/marl_learner.py:193-197); 0 = MSE for the kernels
        self.huber_delta = float(getattr(config, "huber_delta", 1.0) or 0.0) if bool(getattr(config, "use_huber_loss", False)) else 0.0
        assert self.huber_delta >= 0.0, "huber_delta must be positive"
        self.n_actions = model.n_actions
        P = model.params
        self.optimizer = AdamHandle(P, model.trainable_order, self.learning_rate, eps=1e-5,
                                    total_iters=self.total_iters, end_factor=self.end_factor_lr_decay)
        self.scheduler = LinearLRHandle(self.optimizer)
        dev = P.device
        self._cap = 0
        self.sumsq = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.opt_sync = torch.zeros(4 + (P.P + 255) // 256 + 8, dtype=torch.int32, device=dev)   # barrier scratch of xrl_reduce_adam
        self.sums = torch.zeros(8, dtype=torch.float64, device=dev)
        self.sync_replicas_from_rank0()

    def _ensure(self, M):
        if M <= self._cap:
            return
        dev, P = self.model.params.device, self.model.params.P
        self._cap = M
        self.slabs = torch.zeros(32, P, device=dev)
        self.partials = torch.zeros(32, 8, dtype=torch.float64, device=dev)
        self.diag = torch.zeros(2 * M, device=dev)
        xdt = torch.uint8 if getattr(self.model, "obs_shape", None) is not None and len(self.model.obs_shape) == 3 else torch.float32
        self.X = torch.zeros(2 * M, self.model.obs_dim, dtype=xdt, device=dev)
        self.model.plan.ensure(2 * M)
        self.model.target_plan.ensure(M)

    def _as_dev(self, x, dtype=torch.float32):
        return torch.as_tensor(x, device=self.model.params.device).to(dtype).contiguous()

    def _step(self, M, act, rew, ter):
        """self.X rows [0,M) = obs, rows [M,2M) = obs_next (already on the device)."""
        model, opt, A = self.model, self.optimizer, self.n_actions
        S = pick_n_split(M)
        fused = bool(getattr(self.config, "use_fused_q_head", True)) and M <= self.partials.shape[0] and \
            getattr(model, "fused_head", lambda: None)() is not None
        use_tail = fused and bool(getattr(self.config, "use_fused_q_tail", True)) and getattr(model, "fused_tail", lambda: None)() is not None
        q_all, q_next = model.forward_pair(self.X, M, self.double_q, skip_last="tail" if use_tail else fused)   # evalQ (:39) [+ Q_eval(s')], targetQ (:40)
        self._tail_done = False                                       # (set below when the optimiser launch carried the phase's ride-along)
        if use_tail:
            # Basic_CNN + BasicQhead at batch <= 32: everything between the last convolution and the convolution stack's backward
            # pass in one launch (xrl_dqn_tail_td)
            in_slabs = bool(getattr(self.config, "use_tail_slab_gradients", True)) and M <= self.slabs.shape[0] and \
                model.conv.N_SPLIT_IMPLICIT == self.slabs.shape[0]
            model.tail_td(M, self.double_q, act, rew, ter, self.diag, self.partials, self.gamma, slabs=self.slabs if in_slabs else None,
                          huber_delta=self.huber_delta)
            S_opt = model.backward(self.X, M, self.slabs, S, tail=True) or S
            S_loss = M
        elif fused:
            # the Q layer itself, the TD rule and the layer's data gradient: one launch (xrl_dqn_head_td), one partials row per row
            model.head_td(M, self.double_q, act, rew, ter, self.diag, self.partials, self.gamma, huber_delta=self.huber_delta)
            S_opt = model.backward(self.X, M, self.slabs, S, skip_last_dg=True) or S
            S_loss = M
        else:
            d_q = model.d_out
            ops.dqn_td(q_eval=q_all, q_next=q_next, q_next_eval=q_all[M:] if self.double_q else None, actions=act,
                       rewards=rew, terminals=ter, d_q=d_q, diag=self.diag, partials=self.partials, M=M, A=A,
                       ld=q_all.shape[1], n_split=S, gamma=float(self.gamma), dueling=int(getattr(model, "dueling", False)),
                       huber_delta=self.huber_delta)
            S_opt = model.backward(self.X, M, self.slabs, S) or S     # (convolutional nets write 32 row chunks of their own)
            S_loss = S
        P, clip = model.params.P, (self.grad_clip_norm if self.use_grad_clip else 0.0)
        if not self.needs_collective() and self._fused_optimizer_ok(self.gradient_exchange() is not None):
            # slab reduction (+ the average over the ranks, inside the launch) + norm + clip + Adam + LinearLR + periodic
            # hard target update (:50-57) in ONE launch
            # (a one-update phase's draw-counter tick and loss sums ride in the same launch: update_from_buffer)
            ride_kw = self._tail(S_loss) if getattr(self, "_tail", None) else {}      # ride-along of a one-update phase (update_from_buffer)
            conv = getattr(model, "conv", None)
            mirrors, timg = [], None
            if conv is not None and conv.implicit and getattr(self.config, "use_live_weight_images", True):
                # the convolution stack's fragment-ordered weight images follow the step inside this launch (and the target's
                # image the periodic hard update): the passes stop rebuilding them (xrl_gather_images, ~5 us per update / act)
                inv_f, inv_d = conv.inverse_maps()
                img_e, img_t = conv.images(None)[0], conv.images(model.target_flat)[0]
                mirrors, timg = [(inv_f, img_e), (inv_d, img_e)], img_t
            ops.reduce_adam(self.slabs, S_opt, P, model.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip, mirrors,
                            self.opt_sync, target=model.target_flat, target_every=self.sync_frequency,
                            exchange=self.gradient_exchange(), target_image=timg, **ride_kw)
            if mirrors:
                conv.mark_live(model.params.flat, model.target_flat)
            self._tail_done = bool(ride_kw)
            return S_loss
        if getattr(model, "conv", None) is not None:
            model.conv.invalidate()
        ops.grad_reduce(self.slabs, S_opt, P, P, opt.grad, self.sumsq)
        if self.distributed_training and self.world_size > 1:
            from ..dist import allreduce_mean_
            allreduce_mean_(opt.grad)
            ops.grad_reduce(opt.grad, 1, P, P, opt.grad, self.sumsq)
        ops.adam_step(model.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip)
        ops.sync_target(model.params.flat, model.target_flat, P, opt.state, self.sync_frequency)   # :56-57
        return S_loss

    # ------------------------------------------------------------------ whole update phases straight from the HBM replay buffer
    def update_from_buffer(self, memory, n_epochs=1, seed=1, sync=True):
        """`n_epochs` updates (sample -> gather -> forward / TD / backward -> Adam -> target sync) as ONE captured hipGraph
        launch: indices are drawn on the device (xrl_sample_replay_indices follows the filling ring through
        memory.size_dev), the gather writes the uint8 / float32 rows straight into the staging tensor the network reads.
        Same arithmetic as update(**memory.sample()); one host sync per phase -- or none: with `sync=False` (and no user
        callback to serve) the call returns None right after the launch, so the host goes on enqueueing the next vector
        step while the update runs; `flush_info()` later returns the info of the last phase launched."""
        M, dev = memory.batch_size, self.model.params.device
        key = self._phase_key(memory, n_epochs)
        self.ensure_live_images()
        if getattr(self, "_buf_graph_key", None) != key:
            self._ensure(M)
            self._idx = torch.zeros(M, dtype=torch.int64, device=dev)
            # (a re-capture -- a workspace grew behind a get_actions / test() on more rows -- must not restart the replay
            #  draws: the Philox stream is keyed by (seed, epoch, counter), so the counter lives as long as the learner)
            if getattr(self, "_sample_counter", None) is None:
                self._sample_counter = torch.zeros(1, dtype=torch.int32, device=dev)
            self._epoch_sums = torch.zeros(n_epochs, 8, dtype=torch.float64, device=dev)
            self._act, self._rew, self._ter = (torch.zeros(M, device=dev) for _ in range(3))
            dst = {"observations": self.X[:M], "next_observations": self.X[M:2 * M], "actions": self._act,
                   "rewards": self._rew, "terminals": self._ter}
            self._buf_S = pick_n_split(M)

            self._phase_partials = torch.zeros(n_epochs, 32, 8, dtype=torch.float64, device=dev)

            def enqueue():
                # per update: draw, gather, step; the draw counter and the loss sums are settled once per phase
                self._tail_done = False
                for e in range(n_epochs):
                    memory.draw_into(self._idx, dst, seed, e, self._sample_counter)     # draw + gather: one launch
                    self.partials = self._phase_partials[e]
                    if n_epochs == 1:                       # (one update per phase -- the DQN loops: both ride in the optimiser launch)
                        self._tail = lambda S: dict(tick=(self._sample_counter, 1), partials=(self._phase_partials[0], S, self._epoch_sums[0]))
                    try:
                        S = self._step(M, self._act, self._rew, self._ter)
                    finally:
                        self._tail = None
                if not self._tail_done:
                    ops.counter_add(self._sample_counter, n_epochs)
                    ops.sum_partials_batched(self._phase_partials, S, 8, self._epoch_sums, n_epochs, 32 * 8, 8)
            self._buf_enqueue, self._buf_graph, self._buf_graph_key = enqueue, None, key
            enqueue()                                       # this call's phase runs eagerly (lazy allocations happen here) ...
            if not self.needs_collective():
                torch.cuda.synchronize()                    # ... and is then captured for the following calls
                g = ops.Graph()
                with g:
                    enqueue()
                self._buf_graph = g
            self._buf_graph_key = self._phase_key(memory, n_epochs)     # (the workspaces as they are after the eager phase)
        elif self._buf_graph is not None:
            self._buf_graph.launch()
        else:
            self._buf_enqueue()
        self._pending_phase = (n_epochs, M)
        if not sync and isinstance(self.callback, _NullCallback):
            self.iterations += n_epochs
            return None
        return self._phase_info(count=True)

    def _phase_key(self, memory, n_epochs):
        """What a captured update phase is valid for: the buffer, the shape -- and the workspaces it points into (Plan.ensure /
        a larger convolution workspace after an acting call on more rows reallocate them: capture again)."""
        m = self.model
        return (id(memory), n_epochs, memory.batch_size, m.plan.cap, getattr(m, "target_plan", m.plan).cap,
                getattr(getattr(m, "conv", None), "ws_gen", 0))

    def ensure_live_images(self):
        """Before a captured update phase is replayed: the convolution stack's weight images must be the parameters' (the
        captured phase has no xrl_gather_images in it -- the optimiser launch keeps the images current)."""
        conv = getattr(self.model, "conv", None)
        if conv is not None and conv.implicit and getattr(self.config, "use_live_weight_images", True) \
                and getattr(self, "_buf_graph", None) is not None:
            flats = (self.model.params.flat, self.model.target_flat)
            if not all(conv.is_live(f) for f in flats):     # load_state_dict / copy_target / an adopted module wrote parameters
                conv.pack_images([(None, True), (flats[1], False)])
                conv.mark_live(*flats)

    def phase_ready(self, memory, n_epochs):
        """Has update_from_buffer(memory, n_epochs) captured its phase (the launches an enclosing capture may enqueue through
        enqueue_phase)?"""
        return getattr(self, "_buf_graph", None) is not None and getattr(self, "_buf_graph_key", None) == self._phase_key(memory, n_epochs)

    def enqueue_phase(self):
        """The launches of one update phase, for a caller that captures them into a larger graph (DQN_Agent's vector-step pair)."""
        self._buf_enqueue()

    def note_phases(self, memory, n_epochs, k):
        """Host bookkeeping of k phases that ran inside a caller's graph (what update_from_buffer(sync=False) does per call)."""
        self._pending_phase = (n_epochs, memory.batch_size)
        self.iterations += n_epochs * k

    def flush_info(self):
        """Info of the last update phase launched with sync=False ({} if there was none)."""
        return self._phase_info(count=False) if getattr(self, "_pending_phase", None) else {}

    def _phase_info(self, count):
        n_epochs, M = self._pending_phase
        self._pending_phase = None
        sums = self._epoch_sums.cpu().numpy()               # the one host sync of the phase
        st = self.read_optimizer()
        info, A = {}, self.n_actions
        for e in range(n_epochs):
            self.iterations += int(count)
            info = self.callback.on_update_start(self.iterations, policy=self.policy, obs=self.X[:M], act=self._act,
                                                 next_obs=self.X[M:2 * M], rew=self._rew, termination=self._ter) or {}
            info.update({self._key("Qloss"): float(sums[e, 0] / M), self._key("predictQ"): float(sums[e, 1] / M),
                         self._key("learning_rate"): st.last_lr})
            evalQ = self._eval_q(M, A)
            info.update(self.callback.on_update_end(self.iterations, policy=self.policy, info=info, evalQ=evalQ,
                                                    predictQ=self.diag[:M], targetQ=self.diag[M:2 * M],
                                                    loss=info[self._key("Qloss")]) or {})
        return info

    def _eval_q(self, M, A):
        """evalQ of the callback (dqn_learner.py:72): the head output, combined for a dueling head (q_head.py:77)."""
        out = self.model.plan.acts[len(self.model.plan.widths) - 1]
        if getattr(self.model, "dueling", False):
            adv, val = out[:M, :A], out[:M, A:A + 1]
            return val + (adv - adv.mean(dim=-1, keepdim=True))
        return out[:M, :A]

    def update(self, **samples):
        self.iterations += 1
        M = len(samples["obs"])
        self._ensure(M)
        self.X[:M].copy_(torch.as_tensor(samples["obs"], device=self.X.device).reshape(M, -1))
        self.X[M:2 * M].copy_(torch.as_tensor(samples["obs_next"], device=self.X.device).reshape(M, -1))
        act, rew, ter = self._as_dev(samples["actions"]), self._as_dev(samples["rewards"]), self._as_dev(samples["terminals"])
        info = self.callback.on_update_start(self.iterations, policy=self.policy, obs=self.X[:M], act=act,
                                             next_obs=self.X[M:2 * M], rew=rew, termination=ter) or {}
        S = self._step(M, act, rew, ter)
        ops.sum_partials(self.partials, S, 8, self.sums)
        s = self.sums.cpu().numpy()
        st = self.read_optimizer()
        info.update({self._key("Qloss"): float(s[0] / M), self._key("predictQ"): float(s[1] / M),
                     self._key("learning_rate"): st.last_lr})
        A = self.n_actions
        evalQ = self._eval_q(M, A)
        info.update(self.callback.on_update_end(self.iterations, policy=self.policy, info=info, evalQ=evalQ,
                                                predictQ=self.diag[:M], targetQ=self.diag[M:2 * M],
                                                loss=info[self._key("Qloss")]) or {})
        return info


class DuelDQN_Learner(DQN_Learner):
    """Dueling DQN (xuance/torch/learners/qlearning_family/dueldqn_learner.py:13-82): DQN_Learner's update on a network
    with DuelingQValueHead (Q = V + A - mean A, rl_models/heads/q_head.py:42-80).  The head's two streams are two groups of
    the same GEMM launches; the combination and its backward live inside xrl_dqn_td (`dueling = 1`)."""

    def __init__(self, config, model, callback=None):
        super().__init__(config, model, callback)
        assert getattr(self.model, "dueling", False), "DuelDQN_Learner needs a network with DuelingQValueHead (dueling=True)"


class PerDQN_Learner(DQN_Learner):
    """DQN with prioritized replay (xuance/torch/learners/qlearning_family/perdqn_learner.py:13-92): DQN_Learner's update
    (the importance weights of the sample are not used by the reference's loss, :49), returning (|td_error|, info) so that
    the agent can call memory.update_priorities (perdqn_agent.py:46-48).  |td| stays on the device."""

    def update(self, **samples):
        info = super().update(**samples)
        M = int(samples["batch_size"]) if "batch_size" in samples else self.diag.numel() // 2
        td_abs = (self.diag[M:2 * M] - self.diag[:M]).abs()      # td_error = targetQ - predictQ (:48)
        return td_abs, info


class DDQN_Learner(DQN_Learner):
    """Double DQN (xuance/torch/learners/qlearning_family/ddqn_learner.py:13-75): identical to DQN_Learner except that the
    target action is argmax_a Q_eval(s', a) (`targetA = self.model(next_batch).actions`, :40-44); xrl_dqn_td implements
    both rules, the eval network simply runs on 2M rows (obs | next_obs) in the same launches."""

    def __init__(self, config, model, callback=None):
        super().__init__(config, model, callback)
        self.double_q = True

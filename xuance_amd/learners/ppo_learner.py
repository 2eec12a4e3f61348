"""PPO-clip learner on the HIP engine.  Same constructor, ``update(**samples)`` contract, info keys and callback
hooks as xuance/torch/learners/policy_gradient/ppo_learner.py:12-95; the arithmetic runs in
xuance_amd/csrc/{gemm,ppo_loss,optim}.hip through the C ABI (include/xrl_hip.h)."""
import numpy as np
import torch

from .. import ops
from .base import Learner, AdamHandle, LinearLRHandle


def pick_n_split(M):
    """Number of deterministic gradient slabs (= batch chunks of the weight-gradient GEMMs)."""
    return int(min(32, max(1, (M + 255) // 256)))


class PPO_Learner(Learner):
    def __init__(self, config, model, callback=None):
        super().__init__(config, model, callback)
        model = self.model                                          # (a reference nn.Module was adopted by the base class)
        self.vf_coef, self.ent_coef, self.clip_range = config.vf_coef, config.ent_coef, config.clip_range
        P = model.params
        self.optimizer = AdamHandle(P, model.ref_order, self.learning_rate, eps=1e-5, total_iters=self.total_iters,
                                    end_factor=self.end_factor_lr_decay)
        self.scheduler = LinearLRHandle(self.optimizer)
        dev = P.device
        self._cap = 0
        self.sumsq = torch.zeros(1024, dtype=torch.float64, device=dev)
        # everything the host reads back per update phase in ONE copy: 8 loss sums + 4 status words (the whole-rollout
        # launch's time-out / XCC flags, PPO_Agent.persist_status) + 4 words of the fused optimiser's barrier scratch
        self._readback = torch.zeros(12, dtype=torch.float64, device=dev)
        self.sums = self._readback[:8]
        self.status_words = self._readback[8:10].view(torch.int32)
        self.last_status = [0, 0, 0, 0]
        self.keep_diag = True     # keep the per-sample callback tensors (log_prob, ratio, surrogates)
        self.loss_mode = 0        # xrl_ppo_loss_t.mode: 0 PPO-clip, 1 A2C (see A2C_Learner)
        self.sync_replicas_from_rank0()

    def estimate_total_iterations(self):                        # ppo_learner.py:28-33
        buffer_size = self.config.horizon_size * self.config.parallels
        update_times = self.config.running_steps // buffer_size
        return update_times * self.config.n_epochs * self.config.n_minibatch

    def _ensure(self, M):
        self._layered_fused_optimizer()
        if M <= self._cap:
            return
        dev, P = self.model.params.device, self.model.params.P
        self._cap = M
        self.n_split_cap = 32
        self.slabs = torch.zeros(self.n_split_cap, P, device=dev)
        self.partials = torch.zeros(self.n_split_cap, 8, dtype=torch.float64, device=dev)
        self.diag = torch.zeros(4, M, device=dev)
        self.model.plan.ensure(M)

    def _as_dev(self, x, dtype=torch.float32):
        return torch.as_tensor(x, device=self.model.params.device).to(dtype).contiguous()

    # ------------------------------------------------------------------ one minibatch, tensors already on device
    def _step(self, obs, ldx, act, ret, adv, old_logp, M, stats=None, finish=True):
        """Enqueue forward, loss, backward, clip and Adam for one minibatch (no host sync; graph-capturable).
        With finish=False the launches stop after the local gradient reduction (multi-GPU: the caller all-reduces
        the flat gradient, then calls finish_step)."""
        model, opt = self.model, self.optimizer
        S = pick_n_split(M)
        heads = model.forward(obs, M, ldx)
        A = model.action_dim
        d_heads = model.d_heads
        critic = model.head_ld > A                                  # ActorNet (PG) has no value column
        kw = dict(out=heads.data_ptr(), value=heads.data_ptr() + 4 * A if critic else None, actions=act.data_ptr(),
                  adv=None if adv is None else adv.data_ptr(), stats=None if stats is None else stats.data_ptr(),
                  returns=ret.data_ptr(), old_logp=None if old_logp is None else old_logp.data_ptr(),
                  d_out=d_heads.data_ptr(), d_value=d_heads.data_ptr() + 4 * A if critic else None,
                  diag=self.diag.data_ptr() if self.keep_diag else None, partials=self.partials.data_ptr(),
                  M=M, A=A, ld_out=model.head_ld, ld_v=model.head_ld, n_split=S, slab_stride=model.params.P,
                  clip_range=self.clip_range, vf_coef=self.vf_coef, ent_coef=self.ent_coef, mode=self.loss_mode)
        if model.dist == "gaussian":
            ls = getattr(model, "log_std_name", "actor.log_std")
            kw.update(log_std=model.params.ptr(ls),
                      d_log_std=self.slabs.data_ptr() + 4 * model.params.offsets[ls],
                      out_act=ops.ACT[model.activation_action])
        kw.update(getattr(self, "_loss_extra", None) or {})             # (PPOKL_Learner: old distribution + kl_coef)
        ops.ppo_loss(model.dist, **kw)
        if self.loss_mode == 3:                                         # the coefficient schedule, after the loss used it
            ops.ppokl_adapt(self.partials, S, M * (A if model.dist == "gaussian" else 1), self.kl_coef_dev, self.target_kl, self.kl_dev)
        model.backward(obs, M, self.slabs, S, ldx)
        if finish and self._layered_fused_optimizer():
            # slab reduction + norm + clip + Adam (+ the fused kernels' derived layouts when they exist) in ONE launch
            ops.reduce_adam(self.slabs, S, model.params.P, model.params.flat, opt.grad, opt.m, opt.v, model.params.P, opt.state,
                            self._lsumsq, self.grad_clip_norm if self.use_grad_clip else 0.0,
                            self._mirrors if getattr(self, "_mirror", False) else [], self._lsync)
            return S
        ops.grad_reduce(self.slabs, S, model.params.P, model.params.P, opt.grad, self.sumsq)
        if finish:
            if self.distributed_training and self.world_size > 1:
                self.allreduce_grad()
            self.finish_step()
        return S

    def _layered_fused_optimizer(self):
        """May the layered path end in xrl_reduce_adam?  One rank, P % 4 == 0, at most 1 024 blocks of 256 parameters (all
        resident: the launch's barrier spins), not switched off.  Allocates the launch's scratch on first use (outside any
        graph capture: prepare_buffer_update / update() call it before they enqueue)."""
        if not hasattr(self, "_lfo"):
            P = self.model.params.P
            self._lfo = not (self.distributed_training and self.world_size > 1) and ops.reduce_adam_fits(P) \
                and bool(getattr(self.config, "use_fused_optimizer", True))
            if self._lfo:
                dev = self.model.params.device
                self._lsumsq = torch.zeros(1024, dtype=torch.float64, device=dev)
                self._lsync = torch.zeros(4 + (P + 255) // 256 + 8, dtype=torch.int32, device=dev)
        return self._lfo

    def allreduce_grad(self):
        """DDP-equivalent gradient averaging as ONE flat RCCL all-reduce, then the norm of the averaged gradient."""
        from ..dist import allreduce_mean_
        opt, P = self.optimizer, self.model.params.P
        allreduce_mean_(opt.grad)
        ops.grad_reduce(opt.grad, 1, P, P, opt.grad, self.sumsq)

    def allreduce_and_finish(self):
        """Multi-GPU optimiser step: ONE flat RCCL all-reduce (mean) of the gradient, then norm + clip + Adam + derived
        layouts in ONE launch (xrl_reduce_adam over the averaged gradient as a single slab) when the fused kernels are
        in use; the generic two-launch sequence otherwise.  Same numbers either way."""
        from ..dist import allreduce_mean_
        allreduce_mean_(self.optimizer.grad)
        self.finish_after_allreduce()

    def finish_after_allreduce(self):
        """The launches that follow the gradient all-reduce (capturable: no collective in here)."""
        model, opt, P = self.model, self.optimizer, self.model.params.P
        if getattr(self, "_mirror", False) and getattr(self, "opt_sync", None) is not None and self._fused_optimizer_ok(False):
            clip = self.grad_clip_norm if self.use_grad_clip else 0.0
            ops.reduce_adam(opt.grad, 1, P, model.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip,
                            self._mirrors, self.opt_sync)
        else:
            ops.grad_reduce(opt.grad, 1, P, P, opt.grad, self.sumsq)
            self.finish_step()
    def finish_step(self):
        """clip_grad_norm_ + Adam.step + LinearLR.step (ppo_learner.py:63-67).  When the fused kernels are in use the
        same launch also refreshes their derived parameter layouts."""
        model, opt = self.model, self.optimizer
        clip = self.grad_clip_norm if self.use_grad_clip else 0.0
        if getattr(self, "_mirror", False):
            ops.adam_step_mirrors(model.params.flat, opt.grad, opt.m, opt.v, model.params.P, opt.state, self.sumsq, clip,
                                  self._mirrors)
        else:
            ops.adam_step(model.params.flat, opt.grad, opt.m, opt.v, model.params.P, opt.state, self.sumsq, clip)

    def _info(self, M, S, partials=None):
        ops.sum_partials(self.partials if partials is None else partials, S, 8, self.sums)
        sync = getattr(self, "opt_sync", None)
        if sync is None and getattr(self, "_lfo", False):
            sync = self._lsync
        if sync is not None:                                        # xrl_reduce_adam: sync[2] != 0 = barrier time-out
            self._readback[10:12].view(torch.int32).copy_(sync[:4])
        rb = self._readback.cpu().numpy()                           # the one host sync of an update
        s = rb[:8]
        self.last_status = rb[8:10].view(np.int32).tolist()
        self.raise_on_optimizer_timeout(int(rb[10:12].view(np.int32)[2]))
        st = self.optimizer.read()
        return {self._key("actor_loss"): float(-s[0] / M), self._key("critic_loss"): float(s[1] / M),
                self._key("entropy"): float(s[2] / M), self._key("learning_rate"): st.last_lr,
                self._key("predict_value"): float(s[3] / M), self._key("clip_ratio"): float(s[4] / M)}

    # ------------------------------------------------------------------ two-branch Gaussian class D-256-256-{A | 1}: ONE launch
    def wide_eligible(self):
        """xrl_ppo_wide_minibatch (csrc/ppo_wide.hip) covers this learner: PPO-clip loss on the MuJoCo network class
        (configs/ppo/mujoco.yaml:8-13), specialised kernels not switched off (config.use_fused_update / xrl_set_fast_kernels)."""
        return bool(getattr(self.config, "use_fused_update", True)) and self.loss_mode == 0 and type(self).update is PPO_Learner.update \
            and ops.PpoWideState.eligible(self.model) and ops.fast_kernels_enabled()

    def _wide_prepare(self, M):
        """Allocations of the wide path for minibatches of up to M rows (outside any graph capture)."""
        dev, P = self.model.params.device, self.model.params.P
        if getattr(self, "_wide", None) is None:
            self._wide = ops.PpoWideState(self.model)
            self._mirrors = list(self._wide.mirrors)                # the optimiser launch keeps the fragment copy current
            self._mirror = True
            self.opt_sync = torch.zeros(4 + (P + 255) // 256 + 8, dtype=torch.int32, device=dev)
        n_t = (M + 31) // 32
        self._wide.prepare_rows(M)
        if n_t > getattr(self, "_wide_tiles", 0):
            self.fslabs = torch.zeros(n_t, P, device=dev)            # one gradient row per 32-row tile (both branches)
            self.fpartials = torch.zeros(2 * n_t, 8, dtype=torch.float64, device=dev)
            self._wide_tiles, self.slab_stride, self.fold, self.split = n_t, P, None, False
        self._ensure(M)

    def _step_wide(self, obs, act, ret, adv, old_logp, M, stats=None, finish=True, heads=None):
        """One minibatch from staged rows: forward + loss + backward in one launch, then slab reduction + clip + Adam (+ the
        gradient average over the ranks) in a second one.  Pointers may be tensors or raw addresses."""
        m, opt, P = self.model, self.optimizer, self.model.params.P
        n_t = (M + 31) // 32
        dist = self.distributed_training and self.world_size > 1
        xc = self.gradient_exchange() if dist and finish else None
        one_launch_opt = finish and (not dist or xc is not None) and self._fused_optimizer_ok(xc is not None)
        # the middle layers' weight gradient (92 % of the parameters) as a launch of its own over all rows, its parts summed by the
        # optimiser launch (config.use_wide_split_dw1; the two-launch optimiser path keeps the per-tile form)
        split = one_launch_opt and bool(getattr(self.config, "use_wide_split_dw1", True))
        parts = self._wide.launch(M, obs, act, ret, adv, old_logp, self.fslabs, P, self.fpartials, self.clip_range, self.vf_coef,
                                  self.ent_coef, stats=stats, diag=self.diag if self.keep_diag else None, heads=heads, split_dw1=split)
        self._last_S, self._last_partials = 2 * n_t, self.fpartials
        if one_launch_opt:
            clip = self.grad_clip_norm if self.use_grad_clip else 0.0
            ops.reduce_adam(self.fslabs, n_t, P, m.params.flat, opt.grad, opt.m, opt.v, P, opt.state, self.sumsq, clip,
                            self._mirrors, self.opt_sync, exchange=xc, alt=(parts, self._wide.w1_ranges()) if split else None)
            return
        ops.grad_reduce(self.fslabs, n_t, P, P, opt.grad, self.sumsq)
        if finish:
            if dist:
                self.allreduce_grad()
            self.finish_step()

    # ------------------------------------------------------------------ fused path: minibatches straight from HBM
    def prepare_buffer_update(self, memory, bs):
        """Allocate the minibatch staging tensors once (nothing may be allocated while a hipGraph is captured)."""
        if getattr(self, "_stage_bs", 0) == bs:
            return
        dev = self.model.params.device
        self._ensure(bs)
        f = memory.soa
        self._stage = {"observations": torch.zeros((bs,) + tuple(memory.obs_shape), dtype=f.fields["observations"].dtype, device=dev),
                       "actions": torch.zeros((bs,) + tuple(memory.act_shape), device=dev),
                       "returns": torch.zeros(bs, device=dev), "advantages": torch.zeros(bs, device=dev),
                       "aux_old_logp": torch.zeros(bs, device=dev)}
        self.stats = torch.zeros(4096, 2, device=dev)
        self._stage_bs = bs
        self._last_S = pick_n_split(bs)

    # ------------------------------------------------------------------ fused minibatch kernel (xrl_ppo_fused_minibatch)
    def trunk_eligible(self):
        """The shared-trunk family of csrc/ppo_trunk.hip: D -> 128 -> {128 -> A | 128 -> 1} with D <= 24, A <= 8, a categorical or
        Gaussian head (activation_action none / tanh), hidden activation relu / leaky_relu / tanh, the first layer at the front of
        the flat layout -- Basic_MLP [128] + actor [128] + critic [128]: every configs/ppo/classic_control/*.yaml and the MLP ones
        of configs/ppo/box2d/.  PPO-clip loss only (the subclasses with other losses override fused_eligible)."""
        m, plan = self.model, self.model.plan
        D, A = m.obs_dim, m.action_dim
        if not getattr(self.config, "use_fused_update", True) or not ops.fast_kernels_enabled() or self.loss_mode != 0:
            return False
        if list(plan.widths) != [D, 128, 256, A + 1] or not (1 <= D <= 24 and 1 <= A <= 8):
            return False
        if m.activation not in ("relu", "leaky_relu", "tanh") or (m.dist == "gaussian" and m.activation_action not in (None, "tanh")):
            return False
        off = m.params.offsets
        return off.get("representation.model.0.weight", -1) == 0 and off.get("representation.model.0.bias", -1) == 128 * D \
            and m.params.P % 4 == 0

    def cartpole_class(self):
        """(D, A, head) = (4, 2.., categorical): the class with the compile-time kernel instance and the 32-byte transition records."""
        m = self.model
        return m.obs_dim == 4 and m.dist == "categorical"

    def fused_eligible(self, memory):
        """One-launch gather + forward + loss + backward per minibatch: the two-branch Gaussian class (ppo_wide.hip), the
        shared-trunk family (ppo_trunk.hip), or -- categorical head, 4-d observations, middle layers in 32-multiples, a 32-row
        tile's activations + gradients in LDS -- the any-shape kernel (ppo_fused.hip)."""
        m, plan = self.model, self.model.plan
        if self.wide_eligible():                                   # the two-branch Gaussian class has its own kernel
            return tuple(memory.act_shape) == (m.action_dim,)
        if self.trunk_eligible():
            return tuple(memory.act_shape) == (() if m.dist == "categorical" else (m.action_dim,))
        if not getattr(self.config, "use_fused_update", True) or m.dist != "categorical" or m.obs_dim != 4:
            return False
        if len(plan.stages) < 2 or len(plan.stages[0]) != 1 or len(plan.widths) > 6:
            return False
        n_layers = sum(len(s) for s in plan.stages)
        mids = [L for s in plan.stages[1:-1] for L in s]
        if n_layers > 8 or any(L.N % 32 or L.K % 32 for L in mids):
            return False
        ld = lambda w: (w + 7) // 8 * 8 + 4
        floats = sum(2 * 32 * ld(w) for w in plan.widths[1:]) + 8 * 32 * 33 + ops.rollout_cache_floats(plan) + 64
        return floats * 4 <= 150 * 1024 and tuple(memory.act_shape) == ()

    def split_eligible(self, n_tiles):
        """Role-split workgroups -- (tile, branch), csrc/ppo_trunk.hip -- for this minibatch?  Every member of the shared-trunk family,
        at every size: round 3 measured the CartPole class against round 2's single-workgroup-per-tile kernel (ppo_fast_kernel,
        retired): update phase 1.64 vs 2.06 ms at 16 tiles, 1.92 vs 2.23 at 64, 2.27 vs 2.53 at 128, and with 64-row tiles 2.73 vs
        2.87 at 256 (tools/microbench_tiles.py).  config.use_role_split_update: False selects the any-shape kernel (ppo_fused.hip)."""
        return self.trunk_eligible() and bool(getattr(self.config, "use_role_split_update", True))

    def pair_eligible(self, n_tiles):
        """64-row tiles in the role-split kernel (one workgroup per (64-row tile, branch): half the weight stream per CU and half the
        gradient slabs): minibatches of at least 192 32-row tiles on the float32 instruction (measured: 128 tiles 2.40 ms with 64-row
        tiles vs 2.27 with 32-row tiles, 256 tiles 2.73 vs 3.08), of at least 112 where they run the split-product kernel.
        config.use_pair_update: "auto" (default) / True / False."""
        want = getattr(self.config, "use_pair_update", "auto")
        if want == "auto":
            # (round 6: where the 64-row tiles run the split-product kernel they pay from 128 tiles on -- tools/sweep_pair_threshold.py,
            #  update phase of 64 minibatches: 128 tiles 2.27 -> 1.97 ms, 160 tiles 3.06 -> 2.19 ms, 96 tiles a tie at 2.02 ms)
            m = self.model
            bx = bool(getattr(self.config, "use_split_products", True)) and not getattr(self.config, "use_chained_update", False) \
                and ops.split_products_class(m.plan, m.obs_dim, m.action_dim, m.dist)
            want = n_tiles >= (112 if bx else 192)
        else:
            want = bool(want)
        return want and self.split_eligible(n_tiles)

    def prepare_fused(self, memory, bs):
        if getattr(self, "_fused_bs", 0) == bs:
            return
        dev, P = self.model.params.device, self.model.params.P
        if self.wide_eligible():
            assert bs % 4 == 0, "the wide minibatch kernel reads 16-byte aligned row blocks: batch size must be a multiple of 4"
            self._wide_prepare(bs)
            self.n_tiles = (bs + 31) // 32
            self.n_part_rows = 2 * self.n_tiles
            self.stats = torch.zeros(4096, 2, device=dev)
            self._fused_bs = bs
            return
        self._ensure(bs)
        m, D = self.model, self.model.obs_dim
        self.n_tiles = (bs + 31) // 32
        self.split = self.split_eligible(self.n_tiles)                 # role-split workgroups (csrc/ppo_trunk.hip)
        self.pair = self.pair_eligible(self.n_tiles)                   # ... on 64-row tiles
        # (round 4's variant with register-chained forward / backward-data products -- 31.1 us per launch against 25.3 us for
        #  ppo_trunk_kernel, DESIGN.md section 3 -- left the library in round 5: tools/csrc/ppo_chain.hip keeps the source)
        self.records = self.cartpole_class()                           # 32-byte transition records (obs[4] | act | ret | adv | logp)
        # gradient slabs: one per tile; with the role-split kernel a fold region behind the parameters takes the critic
        # role's first-layer gradient, and every (tile, role) workgroup has its own row of loss partials
        l0_fold = 128 * D + 128
        self.slab_stride = P + (l0_fold if self.split else 0)
        self.fold = (P, l0_fold) if self.split else None
        self.n_part_rows = self.n_tiles * (2 if self.split else 1)
        self.n_slabs = (bs + 63) // 64 if self.pair else self.n_tiles  # gradient slabs one minibatch launch writes
        self.fslabs = torch.zeros(self.n_tiles, self.slab_stride, device=dev)
        self.fpartials = torch.zeros(self.n_part_rows, 8, dtype=torch.float64, device=dev)
        self.stats = torch.zeros(4096, 2, device=dev)
        self.sumsq = torch.zeros(256, dtype=torch.float64, device=dev)
        # barrier scratch of xrl_reduce_adam and of the chained minibatch launch (xrl_ppo_trunk_chained)
        self.opt_sync = torch.zeros(max(4 + (P + 255) // 256 + 8, ops.CHAIN_SYNC_WORDS), dtype=torch.int32, device=dev)
        self._fused_bs = bs
        self._pending_opt = None                                       # optimiser step the next minibatch launch will do (chained)
        self._chain_ok = {}
        self._mirrors = []
        self.params_t = self.cache_image = None
        if not self.split or not ops.fast_kernels_enabled():
            # derived layouts of the single-workgroup kernel (ppo_fused: transposed middle weights, packed small parameters).  (Until
            # round 4 the role-split CartPole class kept them current as well, for rollout kernels that have since been replaced: two
            # of the optimiser launch's four mirror maps, ~70 k scattered stores per step, for nobody.  A role-split learner gets
            # them when the specialised kernels are switched off -- here, or on the first such minibatch: _derived_layouts.)
            self._derived_layouts(fill=False)
        nf = ops.mid_frag_floats(self.model.plan)
        self.frag = torch.zeros(nf, device=dev) if nf else None       # MFMA-fragment-ordered copy of the middle layer
        if nf:
            mf, mb = ops.frag_layout_maps(self.model.plan, P, dev)  # forward section: first stream of the minibatch kernel;
            self._mirrors += [(mf, self.frag), (mb, self.frag)]     # backward section: its second one (backward-data)
        # 64-row tiles of the CartPole class: the three 128-wide products as exact 3-way bf16 splits on the matrix cores
        # (csrc/ppo_trunk_bx.hip) -- the branch layer as three bf16 planes, kept current by SPLIT mirror maps of the optimiser launch.
        # config.use_split_products: False keeps the float32 matrix instruction (csrc/ppo_trunk.hip).
        self.frag16 = None
        if nf and self.split and self.pair and getattr(self.config, "use_split_products", True) \
                and ops.split_products_class(self.model.plan, D, self.model.action_dim, self.model.dist) \
                and not getattr(self.config, "use_chained_update", False):
            self.frag16 = torch.zeros(3 * ops.FRAG16_PLANE, dtype=torch.int16, device=dev)
            mf16, mb16 = ops.frag16_layout_maps(self.model.plan, P, dev)
            # (every minibatch launch of this learner then reads the split image: the float32 fragment copy is refreshed once per
            #  update phase -- refresh_fused_params -- and is nobody's operand in between: its two mirror maps leave the optimiser launch)
            self._mirrors = [mm for mm in self._mirrors if mm[1] is not self.frag]
            self._mirrors += [(mf16, self.frag16, ops.FRAG16_PLANE), (mb16, self.frag16, ops.FRAG16_PLANE)]
        self.packed = torch.zeros(memory.n_size * memory.n_envs * 8, device=dev) if self.records else None   # transition records
        self._mirror = True

    def prepare_rows(self, count):
        """Allocate the gathered-record staging of an update phase (outside graph capture)."""
        if getattr(self, "_wide", None) is not None:
            if getattr(self, "_wstage_rows", 0) != count:
                dev, D, A = self.model.params.device, self.model.obs_dim, self.model.action_dim
                self._wstage = {"observations": torch.zeros(count, D, device=dev), "actions": torch.zeros(count, A, device=dev),
                                "returns": torch.zeros(count, device=dev), "advantages": torch.zeros(count, device=dev),
                                "aux_old_logp": torch.zeros(count, device=dev)}
                self._wstage_rows = count
            return
        if getattr(self, "records", True) and (getattr(self, "rows", None) is None or self.rows.numel() != count * 8):
            self.rows = torch.zeros(count * 8, device=self.model.params.device)

    def refresh_fused_params(self, memory=None, idx_all=None):
        """Derived parameter layouts the fused kernel reads (transposed middle weights, packed small parameters); with
        `memory`, also the packed transition records of the finished rollout (once per update phase)."""
        if getattr(self, "_wide", None) is not None:
            self._wide.pack()
            self._rows_idx = None
            if memory is not None and idx_all is not None:
                # every minibatch of the phase gathered with ONE launch into row-contiguous staging
                self.prepare_rows(idx_all.numel())
                st, f = self._wstage, memory.soa
                ops.soa_gather([(st[n], f.fields[n], f.row_bytes[n]) for n in st], idx_all.view(-1), memory.n_envs, memory.n_size)
                self._rows_idx = idx_all
            return
        if memory is not None and getattr(self, "records", True):
            f = memory.soa.fields
            ops.pack_transitions(f["observations"], f["actions"], f["returns"], f["advantages"], f["aux_old_logp"],
                                 self.packed, memory.n_size * memory.n_envs)
            self._packed_valid = True
            self._rows_idx = None
            if idx_all is not None:
                # every minibatch of the phase gathered now: the minibatch kernel's first load is then a contiguous read
                # that needs neither the index nor a second dependent hop
                self.prepare_rows(idx_all.numel())
                ops.gather_rows(self.packed, idx_all, self.rows, idx_all.numel(), memory.n_envs, memory.n_size)
                self._rows_idx = idx_all
        if self.params_t is not None:
            ops.transpose_mid(self.model.plan, self.model.params.flat, self.params_t)
            ops.pack_rollout_cache(self.model.plan, self.model.params.flat, self.cache_image)
        if self.frag is not None:
            ops.pack_mid_frags(self.model.plan, self.model.params.flat, self.frag)
        if getattr(self, "frag16", None) is not None:
            ops.pack_mid_frags16(self.model.plan, self.model.params.flat, self.frag16)

    def _derived_layouts(self, fill=True):
        """params_t / cache_image of the any-shape minibatch kernel (xrl_ppo_fused_minibatch without a fold region needs both) and
        their mirror maps in the optimiser launch; `fill`: from the current parameters (a role-split learner that loses its
        specialised kernels mid-way: ops.set_fast_kernels(False))."""
        if self.params_t is not None:
            return
        P, dev = self.model.params.P, self.model.params.device
        self.params_t = torch.zeros(P, device=dev)
        self.cache_image = torch.zeros(ops.rollout_cache_floats(self.model.plan) + 16, device=dev)
        self.map_t, self.map_img = ops.derived_layout_maps(self.model.plan, P, dev)
        self._mirrors = [(self.map_t, self.params_t), (self.map_img, self.cache_image)] + list(self._mirrors)
        if fill:
            ops.transpose_mid(self.model.plan, self.model.params.flat, self.params_t)
            ops.pack_rollout_cache(self.model.plan, self.model.params.flat, self.cache_image)

    def chain_eligible(self, M, pair):
        """May the optimiser step of a minibatch ride in the NEXT minibatch launch (xrl_ppo_trunk_chained)?  The CartPole class of the
        shared-trunk family on one rank (with several ranks the gradient average sits in the optimiser launch), the one-launch
        optimiser usable, every workgroup of the launch resident and at least ceil(P / 256) of them.  OFF unless
        config.use_chained_update: True -- measured on the headline (round 6, profiles/r06_a_probe_chain.json): 44.4 us per minibatch
        against 42.7 us for the launch pair; the two in-launch barriers cost what the kernel boundary they replace costs
        (DESIGN.md section 3 "Round 6")."""
        key = (M, bool(pair))
        if key not in self._chain_ok:
            ok = bool(getattr(self.config, "use_chained_update", False)) and self.split and ops.fast_kernels_enabled() \
                and self.cartpole_class() and self.model.action_dim == 2 \
                and not (self.distributed_training and self.world_size > 1) and self._fused_optimizer_ok(False) \
                and getattr(self, "frag16", None) is None \
                and ops.ppo_trunk_chain_fits(M, 64 if pair else 32, self.model.params.P)
            self._chain_ok[key] = ok
        return self._chain_ok[key] and ops.fast_kernels_enabled()

    def finish_pending(self):
        """The optimiser step a chained minibatch launch was going to do, as a launch of its own (end of an update phase)."""
        o = getattr(self, "_pending_opt", None)
        if o is not None:
            self._pending_opt = None
            ops.reduce_adam(o["slabs"], o["n_split"], o["slab_stride"], o["params"], o["grad"], o["m"], o["v"], o["P"], o["state"],
                            o["sumsq_part"], o["max_norm"], o["mirrors"], o["sync"], fold=o["fold"])

    def enqueue_minibatch_fused(self, memory, idx, stats=None, finish=True, defer=False):
        """One launch for gather + forward + loss + backward, then reduce + Adam, then refresh the derived layouts.
        defer (the agent's update phase): the optimiser step may wait for the next minibatch launch, which then does it in its
        prologue (xrl_ppo_trunk_chained); finish_pending() after the last minibatch."""
        m, opt, f = self.model, self.optimizer, memory.soa.fields
        M = idx.numel()
        if getattr(self, "_wide", None) is not None:
            base, st = getattr(self, "_rows_idx", None), None
            if base is not None:                           # `idx` is a row of the index matrix the rows were gathered for
                off = (idx.data_ptr() - base.data_ptr()) // 8
                if 0 <= off and off + M <= base.numel() and idx.is_contiguous():
                    st = {n: t[off:off + M] for n, t in self._wstage.items()}
            if st is None:                                 # any other index set: gather it now
                self.prepare_rows(M)
                st, fs = self._wstage, memory.soa
                ops.soa_gather([(st[n], fs.fields[n], fs.row_bytes[n]) for n in st], idx, memory.n_envs, memory.n_size)
            return self._step_wide(st["observations"], st["actions"], st["returns"], st["advantages"], st["aux_old_logp"], M,
                                   stats=stats, finish=finish)
        fold = self.fold if (self.split and ops.fast_kernels_enabled()) else None     # (tests switch the specialised kernels off)
        if fold is None and self.params_t is None:
            self._derived_layouts()                        # (not inside a capture: switch the kernels off before the first update)
        rows = None
        base = getattr(self, "_rows_idx", None)
        if base is not None:                               # `idx` is a row of the index matrix the records were gathered for
            off = (idx.data_ptr() - base.data_ptr()) // 8
            if 0 <= off and off + M <= base.numel() and idx.is_contiguous():
                rows = self.rows[off * 8:(off + M) * 8]
        pair = bool(fold) and getattr(self, "pair", False)
        gauss = m.dist == "gaussian"
        pending = getattr(self, "_pending_opt", None)
        if pending is not None and not (fold and self.chain_eligible(M, pair)):
            self.finish_pending()
            pending = None
        self._pending_opt = None
        launch = ops.ppo_fused_minibatch if pending is None else (lambda plan, **kw: ops.ppo_trunk_chained(plan, pending, **kw))
        launch(m.plan, params=m.params.flat, params_t=self.params_t, cache_image=self.cache_image,
                                f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                                f_logp=f["aux_old_logp"], idx=idx, stats=stats, slabs=self.fslabs, frag_image=self.frag,
                                frag16=self.frag16 if (pair and getattr(self, "frag16", None) is not None) else None,
                                f_packed=self.packed if getattr(self, "_packed_valid", False) else None, f_rows=rows,
                                partials=self.fpartials, diag=self.diag if self.keep_diag else None,
                                slab_stride=self.slab_stride, l0_fold_off=fold[0] if fold else 0, M=M,
                                n_envs=memory.n_envs, T=memory.n_size, D=m.obs_dim, A=m.action_dim, pad0=64 if pair else 0,
                                dist=int(gauss), out_act=ops.ACT[m.activation_action] if gauss else 0,
                                log_std_off=m.params.offsets[getattr(m, "log_std_name", "actor.log_std")] if gauss else 0,
                                clip_range=self.clip_range, vf_coef=self.vf_coef, ent_coef=self.ent_coef,
                                dbg=getattr(self, "_dbg", None))
        n_t = (M + 63) // 64 if pair else (M + 31) // 32             # gradient slabs (and, per role, partial rows) of this launch
        self._last_S, self._last_partials = n_t * (2 if fold else 1), self.fpartials
        dist = self.distributed_training and self.world_size > 1
        xc = self.gradient_exchange() if dist and finish else None
        if finish and (not dist or xc is not None) and self._fused_optimizer_ok(xc is not None):
            # slab reduction (+ with several ranks: the gradient average over the ranks, inside the launch) + clip + Adam +
            # derived layouts in ONE launch (xrl_reduce_adam / xrl_reduce_adam_exchange)
            clip = self.grad_clip_norm if self.use_grad_clip else 0.0
            if defer and fold and self.chain_eligible(M, pair):
                # ... of the NEXT minibatch's launch: its workgroups do this step first (bit-identical; csrc/opt_chain.h)
                self._pending_opt = dict(slabs=self.fslabs, n_split=n_t, slab_stride=self.slab_stride, params=m.params.flat,
                                         grad=opt.grad, m=opt.m, v=opt.v, P=m.params.P, state=opt.state, sumsq_part=self.sumsq,
                                         max_norm=clip, mirrors=self._mirrors, sync=self.opt_sync, fold=fold)
                return
            ops.reduce_adam(self.fslabs, n_t, self.slab_stride, m.params.flat, opt.grad, opt.m, opt.v, m.params.P, opt.state,
                            self.sumsq, clip, self._mirrors, self.opt_sync, fold=fold, exchange=xc)
            return
        ops.grad_reduce(self.fslabs, n_t, self.slab_stride, m.params.P, opt.grad, self.sumsq, fold=fold)
        if finish:
            if dist:
                self.allreduce_grad()
            self.finish_step()

    def enqueue_minibatch_from_buffer(self, memory, idx, stats=None, finish=True):
        """memory.sample(idx) + update(**samples) without materialising Python objects: one gather launch
        (advantages normalised on the fly, memory_tools.py:281-282) followed by the update launches."""
        st, f = self._stage, memory.soa
        names = list(st)
        flags = [1 if (n == "advantages" and stats is not None) else 0 for n in names]
        ops.soa_gather([(st[n], f.fields[n], f.row_bytes[n]) for n in names], idx, memory.n_envs, memory.n_size,
                       stats=stats, flags=flags)
        M = idx.numel()                                     # (a short final minibatch uses the first M staging rows)
        obs = st["observations"][:M].view(M, -1)
        self._last_S = self._step(obs, obs.shape[1], st["actions"], st["returns"], st["advantages"], st["aux_old_logp"], M,
                                  finish=finish)
        self._last_partials = self.partials

    def last_info(self, M):
        """Info dict of the most recent minibatch (what train_epochs returns, on_policy.py:205)."""
        return self._info(M, self._last_S, getattr(self, "_last_partials", None))

    # ------------------------------------------------------------------ reference API (ppo_learner.py:35-95)
    def update(self, **samples):
        self.iterations += 1
        # (uint8 frame stacks stay uint8: x / 255 happens in the first convolution's im2col, cnn.py:99)
        obs = self._as_dev(samples["obs"], torch.uint8 if getattr(self.model, "obs_shape", None) is not None else torch.float32)
        act = self._as_dev(samples["actions"])
        ret = self._as_dev(samples["returns"])
        adv = self._as_dev(samples["advantages"])
        old_logp = self._as_dev(samples["aux_batch"]["old_logp"])
        M = obs.shape[0]
        obs = obs.reshape(M, -1)
        self._ensure(M)
        info = self.callback.on_update_start(self.iterations, policy=self.policy, obs=obs, act=act, returns=ret,
                                             advantages=adv, old_logp=old_logp) or {}
        A = self.model.action_dim
        if self.wide_eligible():
            self._wide_prepare(M)
            self._wide.pack()                                       # (the parameters may have been loaded since the last call)
            if getattr(self, "_wheads", None) is None or self._wheads.shape[0] < M:
                self._wheads = torch.zeros(M, A + 1, device=obs.device)
            heads = self._wheads
            self._step_wide(obs, act.reshape(M, -1).contiguous(), ret, adv, old_logp, M, heads=heads)
            info.update(self._info(M, self._last_S, self.fpartials))
        else:
            S = self._step(obs, obs.shape[1], act, ret, adv, old_logp, M)
            info.update(self._info(M, S))
            heads = self.model.plan.acts[len(self.model.plan.widths) - 1]
        d = self.diag.view(-1)                                      # the loss kernel packs [4][M] for the current M
        cb = dict(policy=self.policy, info=info, v_pred=heads[:M, A], log_prob=d[0:M], ratio=d[M:2 * M],
                  surrogate1=d[2 * M:3 * M], surrogate2=d[3 * M:4 * M],
                  a_loss=info[self._key("actor_loss")], c_loss=info[self._key("critic_loss")],
                  e_loss=info[self._key("entropy")])
        cb["loss"] = cb["a_loss"] - self.ent_coef * cb["e_loss"] + self.vf_coef * cb["c_loss"]
        cb["a_dist"] = heads[:M, :A]
        info.update(self.callback.on_update_end(self.iterations, **cb) or {})
        return info


class A2C_Learner(PPO_Learner):
    """Advantage actor-critic (xuance/torch/learners/policy_gradient/a2c_learner.py:11-90): the PPO machinery with the
    actor term -(adv * log_prob).mean() instead of the clipped surrogate (xrl_ppo_loss_t.mode = 1), no old log-prob, the
    reference's info keys (`actor-loss`, `critic-loss`, ...) and its LinearLR horizon (total_iters = running_steps, :19-21).
    Always the layered path (the fused minibatch kernel implements the PPO-clip loss only)."""

    def __init__(self, config, model, callback=None):
        if not hasattr(config, "clip_range"):
            config.clip_range = 0.0
        super().__init__(config, model, callback)
        self.loss_mode = 1

    def estimate_total_iterations(self):                        # a2c_learner.py:19-21: total_iters=self.config.running_steps
        return int(self.config.running_steps)

    def fused_eligible(self, memory):
        return False

    def _info(self, M, S, partials=None):
        i = super()._info(M, S, partials)
        k = self._key
        return {k("actor-loss"): i[k("actor_loss")], k("critic-loss"): i[k("critic_loss")], k("entropy"): i[k("entropy")],
                k("learning_rate"): i[k("learning_rate")], k("predict_value"): i[k("predict_value")]}

    def update(self, **samples):                                  # a2c_learner.py:34-90
        self.iterations += 1
        obs = self._as_dev(samples["obs"])
        act, ret, adv = self._as_dev(samples["actions"]), self._as_dev(samples["returns"]), self._as_dev(samples["advantages"])
        M = obs.shape[0]
        obs = obs.reshape(M, -1)
        self._ensure(M)
        info = self.callback.on_update_start(self.iterations, model=self.policy, obs=obs, act=act, returns=ret,
                                             advantages=adv) or {}
        S = self._step(obs, obs.shape[1], act, ret, adv, adv, M)     # (old_logp is not read in mode 1)
        info.update(self._info(M, S))
        heads = self.model.plan.acts[len(self.model.plan.widths) - 1]
        A = self.model.action_dim
        cb = dict(model=self.policy, info=info, v_pred=heads[:M, A], log_prob=self.diag.view(-1)[0:M],
                  a_loss=info[self._key("actor-loss")], c_loss=info[self._key("critic-loss")], e_loss=info[self._key("entropy")])
        cb["loss"] = cb["a_loss"] - self.ent_coef * cb["e_loss"] + self.vf_coef * cb["c_loss"]
        info.update(self.callback.on_update_end(self.iterations, **cb) or {})
        return info


class PPOKL_Learner(PPO_Learner):
    """PPO with a KL penalty (xuance/torch/learners/policy_gradient/ppokl_learner.py:14-101): actor term
    -(ratio * adv).mean() + kl_coef * KL(new || old).mean() with the old log-prob AND the KL taken from the old distribution's
    parameters (`aux_batch["old_dist"]`), no clipping; kl_coef doubles / halves around `target_kl` after every update and
    stays in [0.1, 20] (:62-66) -- on the device (xrl_ppokl_adapt), so chained updates need no host round trip.  The
    reference's own update raises AttributeError as shipped (:48 reads `model_output.distribution`); the fixtures run it
    unmodified on a model whose output carries that attribute name (oracle/make_golden.py: golden_ppokl).  Torch's
    Normal-Normal KL is elementwise, so for Gaussian policies `kl.mean()` averages over rows x action dims; kept.
    `old_dist` may be the reference's array / list of per-sample distribution objects (split_distributions), or
    {"logits": [M, A]} / {"mu": [M, A], "std": [M, A] or [A]} arrays.  Layered path (xrl_ppo_loss_t.mode = 3)."""

    def __init__(self, config, model, callback=None):
        if not hasattr(config, "clip_range"):
            config.clip_range = 0.0
        super().__init__(config, model, callback)
        self.loss_mode = 3
        dev = self.model.params.device
        self.target_kl = float(config.target_kl)
        self.kl_coef = float(config.kl_coef)                      # host mirror of the device value (read back with the losses)
        self.kl_coef_dev = torch.tensor([self.kl_coef, self.kl_coef], dtype=torch.float64, device=dev)   # [current, used by the last loss]
        self.kl_dev = torch.zeros(1, device=dev)

    def fused_eligible(self, memory):
        return False

    def prepare_buffer_update(self, memory, bs):               # the agent loop: the old distribution comes from the buffer's
        fresh = getattr(self, "_stage_bs", 0) != bs               # aux fields old_a (logits / mu) and old_b (std)
        super().prepare_buffer_update(memory, bs)
        if fresh:
            dev, A = self.model.params.device, self.model.action_dim
            self._stage["aux_old_a"] = torch.zeros(bs, A, device=dev)
            if self.model.dist == "gaussian":
                self._stage["aux_old_b"] = torch.zeros(bs, A, device=dev)
            self._loss_extra = dict(old_a=self._stage["aux_old_a"].data_ptr(),
                                    old_b=self._stage["aux_old_b"].data_ptr() if "aux_old_b" in self._stage else None,
                                    kl_coef=self.kl_coef_dev.data_ptr())

    def old_dist_arrays(self, od, M):
        """-> (old_a [M, A], old_b [M, A] or None) device tensors from whatever `aux_batch['old_dist']` holds."""
        A, dev = self.model.action_dim, self.model.params.device
        if isinstance(od, dict):
            if "logits" in od:
                return self._as_dev(od["logits"]).reshape(M, A), None
            std = self._as_dev(od["std"])
            return self._as_dev(od["mu"]).reshape(M, A), (std.expand(M, A) if std.dim() == 1 else std.reshape(M, A)).contiguous()
        items = list(np.asarray(od, dtype=object).reshape(-1))    # the reference's per-sample distribution objects
        if hasattr(items[0], "logits") and items[0].logits is not None:
            return torch.cat([torch.as_tensor(d.logits).reshape(1, A) for d in items]).to(dev).float().contiguous(), None
        mu = torch.stack([torch.as_tensor(d.mu).reshape(A) for d in items]).to(dev).float().contiguous()
        std = torch.stack([torch.as_tensor(d.std).reshape(A) for d in items]).to(dev).float().contiguous()
        return mu, std

    def _info(self, M, S, partials=None):
        ops.sum_partials(self.partials if partials is None else partials, S, 8, self.sums)
        rb = torch.cat([self.sums, self.kl_coef_dev]).cpu().numpy()   # the one host sync of an update
        s = rb[:8]
        # the coefficient the LAST update's loss was formed with comes from the device (xrl_ppokl_adapt stores it): the host
        # mirror is stale after a chain of captured updates
        self.kl_coef, used = float(rb[8]), float(rb[9])
        count = M * (self.model.action_dim if self.model.dist == "gaussian" else 1)
        kl = float(np.float32(s[5] / count))
        k, st = self._key, self.read_optimizer()
        return {k("actor-loss"): float(-s[0] / M + used * kl), k("critic-loss"): float(s[1] / M), k("entropy"): float(s[2] / M),
                k("learning_rate"): st.last_lr, k("kl"): kl, k("predict_value"): float(s[3] / M)}

    def update(self, **samples):                                  # ppokl_learner.py:35-101
        self.iterations += 1
        obs = self._as_dev(samples["obs"])
        act, ret, adv = self._as_dev(samples["actions"]), self._as_dev(samples["returns"]), self._as_dev(samples["advantages"])
        M = obs.shape[0]
        obs = obs.reshape(M, -1)
        self._ensure(M)
        old_dists = samples["aux_batch"]["old_dist"]
        info = self.callback.on_update_start(self.iterations, model=self.policy, obs=obs, act=act, returns=ret,
                                             advantages=adv, old_dists=old_dists) or {}
        old_a, old_b = self.old_dist_arrays(old_dists, M)
        self._loss_extra = dict(old_a=old_a.data_ptr(), old_b=None if old_b is None else old_b.data_ptr(),
                                kl_coef=self.kl_coef_dev.data_ptr())
        S = self._step(obs, obs.shape[1], act, ret, adv, None, M)
        info.update(self._info(M, S))
        heads = self.model.plan.acts[len(self.model.plan.widths) - 1]
        A, k = self.model.action_dim, self._key
        d = self.diag.view(-1)
        cb = dict(model=self.policy, info=info, a_dist=heads[:M, :A], v_pred=heads[:M, A], log_prob=d[0:M], ratio=d[M:2 * M],
                  kl=info[k("kl")], a_loss=info[k("actor-loss")], c_loss=info[k("critic-loss")], e_loss=info[k("entropy")])
        cb["loss"] = cb["a_loss"] - self.ent_coef * cb["e_loss"] + self.vf_coef * cb["c_loss"]
        info.update(self.callback.on_update_end(self.iterations, **cb) or {})
        return info


class PG_Learner(PPO_Learner):
    """Vanilla policy gradient (xuance/torch/learners/policy_gradient/pg_learner.py:10-77): a_loss = -(returns * log_prob).mean()
    with the entropy bonus, on an actor-only model (nets.ActorNet = VanillaPolicyGradient) -- `xrl_ppo_loss_t.mode = 2`, no
    critic columns, the reference's info keys (`actor-loss`, `entropy`, `learning_rate`)."""

    def __init__(self, config, model, callback=None):
        for k, v in (("clip_range", 0.0), ("vf_coef", 0.0)):
            if not hasattr(config, k):
                setattr(config, k, v)
        super().__init__(config, model, callback)
        self.loss_mode = 2

    def fused_eligible(self, memory):
        return False

    def _info(self, M, S, partials=None):
        i = super()._info(M, S, partials)
        k = self._key
        return {k("actor-loss"): i[k("actor_loss")], k("entropy"): i[k("entropy")], k("learning_rate"): i[k("learning_rate")]}

    def update(self, **samples):                                  # pg_learner.py:30-71
        self.iterations += 1
        obs, act, ret = self._as_dev(samples["obs"]), self._as_dev(samples["actions"]), self._as_dev(samples["returns"])
        M = obs.shape[0]
        obs = obs.reshape(M, -1)
        self._ensure(M)
        info = self.callback.on_update_start(self.iterations, model=self.policy, obs=obs, act=act, returns=ret) or {}
        S = self._step(obs, obs.shape[1], act, ret, None, None, M)
        info.update(self._info(M, S))
        a_loss, e_loss = info[self._key("actor-loss")], info[self._key("entropy")]
        info.update(self.callback.on_update_end(self.iterations, model=self.policy, info=info, log_prob=self.diag.view(-1)[0:M],
                                                a_loss=a_loss, e_loss=e_loss, loss=a_loss - self.ent_coef * e_loss) or {})
        return info

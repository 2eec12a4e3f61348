"""PPO-clip agent: rollout + update loop on the HIP engine.

Mirrors xuance/torch/agents/policy_gradient/ppo_agent.py:12-181 with core/on_policy.py:23-300 (constructor
signature ``(config, envs, callback)``, ``train(train_steps)``, ``_build_model/_build_memory/_build_learner``,
``get_actions``, obs/reward normalisation, per-env path closing), restructured for the device:

  * one vector step = 6 small launches (normalise+store, 3 grouped GEMMs, sample+store, env, bookkeeping);
    the policy batch carries 2n rows -- the n current observations and the n (normalised) next observations of
    the previous step -- so the bootstrap value V(next_obs) the reference obtains with extra forward passes
    (ppo_agent.py:130,156) comes out of the same launch;
  * a whole rollout (T steps) and a whole update phase (n_epochs x n_minibatch minibatches) are each captured
    into one hipGraph and replayed;
  * with more than one rank the envs are sharded and gradients are all-reduced (xuance_amd/dist.py).
"""
from argparse import Namespace

import numpy as np
import torch

from .. import ops
from .base import AgentSurface, ActionOutput
from ..learners.ppo_learner import PPO_Learner
from ..memory import HipOnPolicyBuffer
from ..nets import ActorCriticNet
from ..spaces import is_discrete, space2shape


def _get(cfg, name, default=None):
    return getattr(cfg, name, default)


class PPO_Agent(AgentSurface):
    def __init__(self, config: Namespace, envs, callback=None):
        self.config, self.envs, self.callback = config, envs, callback
        self.device = _get(config, "device", "cuda")
        self._init_surface()
        self.n_envs = envs.num_envs
        self.observation_space, self.action_space = envs.observation_space, envs.action_space
        self.gamma = config.gamma
        self.gae_lam = _get(config, "gae_lambda", 0.95)
        self.horizon_size, self.n_epochs, self.n_minibatch = config.horizon_size, config.n_epochs, config.n_minibatch
        self.use_obsnorm, self.use_rewnorm = _get(config, "use_obsnorm", False), _get(config, "use_rewnorm", False)
        self.obsnorm_range, self.rewnorm_range = _get(config, "obsnorm_range", 5.0), _get(config, "rewnorm_range", 5.0)
        self.seed = int(_get(config, "seed", 1))
        self.current_step = 0
        self.use_graph = _get(config, "use_hip_graph", True)
        dev, n = self.device, self.n_envs
        self.obs_dim = int(np.prod(space2shape(self.observation_space)))
        # image observations (configs/ppo/atari.yaml: 84 x 84 x 4 uint8 frame stacks): AC_CNN_Atari network, uint8 buffer
        self.frames = len(space2shape(self.observation_space)) == 3
        self.model = self._build_model()
        self.memory = self._build_memory(self.auxiliary_info_shape)
        self.learner = self._build_learner(self.config, self.model, self.callback)
        # running statistics (statistic_tools.py:65-110: mean 0, var 1, count 1e-4)
        D = self.obs_dim
        self.obs_mean = torch.zeros(D, device=dev)
        self.obs_var = torch.ones(D, device=dev)
        self.obs_count = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
        self.ret_mean = torch.zeros(1, device=dev)
        self.ret_var = torch.ones(1, device=dev)
        self.ret_count = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
        self.returns = torch.zeros(n, device=dev)               # discounted return tracker (ppo_agent.py:144)
        self.X = torch.zeros(2 * n, D, device=dev) if not self.frames else None   # policy input: [obs_t ; next_obs_{t-1}] (normalised)
        self.Xu8 = torch.zeros(2 * n, D, dtype=torch.uint8, device=dev) if self.frames else None   # the same rows as raw frames
        # a provider that can write into the policy's input batches (envs/synthetic.py: bind_policy_batch) saves the two frame copies of a
        # vector step: two batches alternate with the provider's observation buffers
        self._acting_fast = False                                   # (set while a rollout is enqueued: _enqueue_rollout)
        self._xin = None
        if self.frames and hasattr(self.envs, "bind_policy_batch") and bool(_get(config, "use_bound_frames", True)):
            self._xin = [torch.zeros(2 * n, D, dtype=torch.uint8, device=dev) for _ in range(2)]
            self.envs.bind_policy_batch(self._xin)
        if self.frames:
            assert not self.use_obsnorm, "uint8 frames are stored and fed as they are (configs/ppo/atari.yaml: use_obsnorm False)"
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)   # RNG counter base, advanced per rollout
        self.perm_counter = torch.zeros(1, dtype=torch.int32, device=dev)   # one tick per update phase (index generation)
        self.buffer_size = n * self.horizon_size
        self.batch_size = self.buffer_size // self.n_minibatch
        # the dense workspaces at their FINAL size before anything is captured: Plan.ensure reallocates when it grows, and a
        # rollout graph captured at 2 n rows would keep writing to the old (freed, soon reused) activations once the learner's
        # first update had asked for minibatch rows
        self.model.plan.ensure(max(2 * n, self.batch_size, self.buffer_size - self.n_minibatch * self.batch_size))
        self.idx = torch.zeros(self.n_epochs * self.n_minibatch, self.batch_size, dtype=torch.int64, device=dev)
        # buffer_size not divisible by n_minibatch: the reference's loop `range(0, buffer_size, batch_size)` ends every epoch
        # with one SHORT minibatch of the remaining transitions (on_policy.py:198-203)
        self.rem = self.buffer_size - self.n_minibatch * self.batch_size
        self.idx_full = torch.zeros(self.n_epochs, self.buffer_size, dtype=torch.int64, device=dev) if self.rem else None
        self._rollout_graph = None
        self._update_graph = None
        self._started = False
        self._act_calls = 1 << 20                                   # Philox step offset of get_actions() draws (outside the rollout's range)
        # Supplied randomness for the rollout's action draws (xrl_sample_t.noise), [horizon_size, n] uniforms (categorical) or
        # [horizon_size, n, A] standard normals (Gaussian), refilled by the caller before every rollout: replays of recorded runs
        # (set_action_noise; layered rollout only -- the one-launch rollout kernels draw from their Philox streams)
        self.action_noise = None
        # fused rollout (one launch per vector step) for the device CartPole with a categorical policy
        from ..envs.cartpole import DeviceCartPoleVecEnv
        self.use_fused_rollout = bool(_get(config, "use_fused_rollout", True)) and self.model.dist == "categorical" and \
            (isinstance(envs, DeviceCartPoleVecEnv) or getattr(envs, "is_cartpole_tape", False)) and self.horizon_size % 2 == 0
        if self.use_fused_rollout:
            # ping-pong copies of everything one workgroup reads from another workgroup's previous step
            self.pp = {"obs_raw": torch.zeros(2, n, D, device=dev), "xnext": torch.zeros(2, n, D, device=dev),
                       "obs_stats": torch.stack([torch.cat([self.obs_mean, self.obs_var])] * 2).contiguous(),
                       "obs_count": torch.full((2, 1), 1e-4, dtype=torch.float64, device=dev),
                       "ret_stats": torch.tensor([[0.0, 1.0], [0.0, 1.0]], device=dev),
                       "ret_count": torch.full((2, 1), 1e-4, dtype=torch.float64, device=dev),
                       "ended": torch.zeros(2, (n + 3) // 4 * 4, dtype=torch.uint8, device=dev),   # rows word-aligned
                       "ret_final": torch.zeros(2, n, device=dev)}
            self.cache_image = torch.zeros(ops.rollout_cache_floats(self.model.plan) + 16, device=dev)
            self.frag_image = torch.zeros(self.model.params.P, device=dev)
        if torch.cuda.is_available():
            ops.init_device()

    # -- builders (same hooks as the reference) -----------------------------------------------------------
    @property
    def auxiliary_info_shape(self):
        return {"old_logp": ()}

    def _build_model(self):
        c = self.config
        discrete = is_discrete(self.action_space)
        if self.frames:                                            # representation: "AC_CNN_Atari" (cnn.py:53-102)
            from ..nets import ActorCriticCNN
            assert discrete, "the convolutional actor-critic has a categorical head (policy: Categorical_AC)"
            return ActorCriticCNN(tuple(space2shape(self.observation_space)), self.action_space.n, tuple(_get(c, "kernels", (8, 4, 3))),
                                  tuple(_get(c, "strides", (4, 2, 1))), tuple(_get(c, "filters", (32, 64, 64))),
                                  tuple(_get(c, "fc_hidden_sizes", (512,))), tuple(_get(c, "actor_hidden_size", ()) or ()),
                                  tuple(_get(c, "critic_hidden_size", ()) or ()), _get(c, "activation", "relu"), device=self.device)
        rep = list(_get(c, "representation_hidden_size", []) or []) if _get(c, "representation", "Basic_MLP") != "Basic_Identical" else []
        return ActorCriticNet(self.obs_dim, self.action_space.n if discrete else int(self.action_space.shape[0]),
                              "categorical" if discrete else "gaussian", rep, list(c.actor_hidden_size),
                              list(c.critic_hidden_size), _get(c, "activation", "leaky_relu"),
                              None if discrete else _get(c, "activation_action", "tanh"), device=self.device)

    def _build_memory(self, auxiliary_info_shape=None):
        c = self.config
        if self.frames:                                            # DummyOnPolicyBuffer_Atari (memory_tools.py:290-328): uint8 frames
            from ..memory import HipOnPolicyBuffer_Atari
            return HipOnPolicyBuffer_Atari(self.observation_space, self.action_space, auxiliary_info_shape, self.n_envs,
                                           self.horizon_size, _get(c, "use_gae", True), _get(c, "use_advnorm", True), self.gamma,
                                           self.gae_lam, device=self.device)
        return HipOnPolicyBuffer(self.observation_space, self.action_space, auxiliary_info_shape, self.n_envs,
                                 self.horizon_size, _get(c, "use_gae", True), _get(c, "use_advnorm", True), self.gamma,
                                 self.gae_lam, device=self.device)

    def _build_learner(self, *args):
        return PPO_Learner(*args)

    # -- one vector step on the device ------------------------------------------------------------------------
    def _enqueue_step_frames(self, t):
        """One vector step on uint8 frame stacks: the frames go to the buffer slot and to the network as they are (x / 255 is
        the first convolution's im2col), rows [n, 2n) of the policy batch are the previous step's next frames (their values
        bootstrap truncated paths, ppo_agent.py:130,156)."""
        env, n, A, f = self.envs, self.n_envs, self.model.action_dim, self.memory.soa.fields
        cur = env.buf_obs.view(n, -1)
        if self._frame_tail():
            # (round 6) everything behind the hidden layer's product as ONE launch (xrl_ppo_act_tail): its split-K epilogue, logits + value,
            # sample / log-prob / value / bootstrap value, the PREVIOUS step's bookkeeping and the copy of `cur` into the buffer slot --
            # five small launches of a vector step (the last step's bookkeeping follows the loop: _enqueue_rollout_tail)
            x = self._xin[env._cur]
            self.model.act_tail(x, 2 * n, n, post=self._post_args(t - 1, (self.obs_mean, self.obs_var, self.obs_count), None) if t > 0 else None,
                                copy=(cur, f["observations"][t], cur.numel() * cur.element_size()),
                                noise=None if self.action_noise is None else self.action_noise[t], act_out=f["actions"][t],
                                val_out=f["values"][t], logp_out=f["aux_old_logp"][t], env_action=env.action,
                                bootv_prev=f["bootv"][t - 1] if t > 0 else None, seed=self.seed, step=t, step_dev=self.step_counter)
            env.step_device(offset=t)
            return
        f["observations"][t].view(n, -1).copy_(cur)               # memory.observations[t] = obs (uint8, ppo_agent.py:128)
        if self._xin is not None:
            heads = self.model.forward(self._xin[env._cur], 2 * n, keep=False, acting=self._acting_fast)   # [obs_t ; next_obs_{t-1}], written there by the provider
        else:
            self.Xu8[:n].copy_(cur)
            heads = self.model.forward(self.Xu8, 2 * n, keep=False, acting=self._acting_fast)
        ops.policy_sample(heads=heads, log_std=None, noise=None if self.action_noise is None else self.action_noise[t],
                          act_out=f["actions"][t], val_out=f["values"][t], logp_out=f["aux_old_logp"][t],
                          env_action=env.action, env_action_f=None, bootv_prev=f["bootv"][t - 1] if t > 0 else None, n=n, A=A,
                          ld=A + 1, gaussian=0, seed=self.seed, step=t, step_dev=self.step_counter)
        if hasattr(env, "advance"):
            env.step_device(offset=t)                               # static step index: the env's counter ticks once per rollout
        else:
            env.step_device()
        if self._xin is None:
            self.Xu8[n:].copy_(env.next_obs.view(n, -1))
        ops.rollout_poststep(**self._post_args(t, (self.obs_mean, self.obs_var, self.obs_count), None))

    def _frame_tail(self):
        """May a vector step on frame stacks end in xrl_ppo_act_tail?  The fast acting pass is on (a rollout is being enqueued), the
        provider writes the policy's batches and takes static step offsets, the network is the class the launch serves, nobody wants a
        callback between the steps; config.use_frame_act_tail: False keeps the launches."""
        if not (self._acting_fast and self._xin is not None and hasattr(self.envs, "advance")) or self._per_step():
            return False
        if not hasattr(self, "_ftail_ok"):
            self._ftail_ok = bool(_get(self.config, "use_frame_act_tail", True)) and type(self)._enqueue_step is PPO_Agent._enqueue_step and \
                hasattr(self.model, "act_tail_eligible") and self.model.act_tail_eligible(2 * self.n_envs) and \
                (self.n_envs * int(np.prod(self.envs.buf_obs.shape[1:]))) % 16 == 0
        return self._ftail_ok

    def _policy_frames(self):
        """The policy's uint8 input batch [obs ; previous next_obs] as it stands now."""
        return self._xin[self.envs._cur] if self._xin is not None else self.Xu8

    def _trunk_forward(self):
        """The acting pass of the general path as ONE launch (xrl_trunk_forward16: csrc/ppo_trunk_bx.hip's forward-only instances) instead
        of the three of the layered forward?  A shared-trunk network with D <= 8, A <= 4 (every classic-control / LunarLander PPO yaml)
        (the reference's default sizes included): the three-plane bf16 image of the branch layer -- the learner's where its update phase
        runs the split-product minibatch kernel, else the agent's own -- is re-packed once at the start of every rollout (one 3 us
        launch: parameters loaded from outside are never stale).  config.use_trunk_forward: False keeps the layered forward.  Returns the image or None."""
        if not hasattr(self, "_tf16"):
            self._tf16 = None
            lr, m = self.learner, self.model
            plan = getattr(m, "plan", None)
            ok = bool(_get(self.config, "use_trunk_forward", True)) and not self.frames and plan is not None \
                and type(self)._enqueue_step is PPO_Agent._enqueue_step and ops.fast_kernels_enabled() \
                and hasattr(m, "dist") and ops.split_products_class(plan, m.obs_dim, m.action_dim, m.dist) \
                and getattr(m, "activation", None) in ("relu", "leaky_relu", "tanh") \
                and (m.dist != "gaussian" or getattr(m, "activation_action", None) in (None, "tanh")) \
                and not self.use_fused_rollout and self._wide_acting() is None
            if ok:
                # (any learner of the PPO family -- PPO-clip, PPO-KL, A2C: the acting pass does not depend on the loss)
                if hasattr(lr, "trunk_eligible") and lr.trunk_eligible() and lr.fused_eligible(self.memory):
                    lr.prepare_fused(self.memory, self.batch_size)    # (outside any capture: _launch_rollout asks before it captures)
                plan.ensure(2 * self.n_envs)
                # the learner's image where its update phase runs the split-product kernel (minibatches of >= 112 tiles), else one of the
                # agent's own: either way re-packed at the start of every rollout (_enqueue_rollout)
                self._tf16 = lr.frag16 if getattr(lr, "frag16", None) is not None else \
                    torch.zeros(3 * ops.FRAG16_PLANE, dtype=torch.int16, device=self.device)
        return self._tf16

    def _acting_forward(self, rows):
        """Head buffer [rows][A + 1] of the acting pass over self.X: one launch for the shared-trunk class, the layered forward else."""
        img = self._trunk_forward()
        m = self.model
        if img is None or rows > m.plan.cap:
            return m.forward(self.X, rows)
        heads = m.plan.acts[len(m.plan.widths) - 1]
        gauss = m.dist == "gaussian"
        ops.trunk_forward16(m.plan, m.params.flat, img, self.X, rows, heads, m.plan.widths[-1], m.obs_dim, m.action_dim, gauss,
                            ops.ACT[m.activation_action] if gauss else 0)
        return heads

    def _act_sample(self, rows, **kw):
        """Acting pass over self.X[:rows] + xrl_policy_sample's work (kw: its arguments without `heads`): ONE launch for the shared-trunk
        class (xrl_trunk_forward16 with a sample argument: the actor role's workgroups sample, the critic role's write the values; no
        head buffer in between; config.use_trunk_sample: False keeps the two launches), else forward + xrl_policy_sample."""
        img, m = self._trunk_forward(), self.model
        if img is not None and rows <= m.plan.cap and bool(_get(self.config, "use_trunk_sample", True)):
            gauss = m.dist == "gaussian"
            ops.trunk_forward16(m.plan, m.params.flat, img, self.X, rows, None, 0, m.obs_dim, m.action_dim, gauss,
                                ops.ACT[m.activation_action] if gauss else 0, sample=kw)
        else:
            ops.policy_sample(heads=self._acting_forward(rows), **kw)

    def _device_tail(self):
        """May a vector step of the general path end in xrl_act_tail (heads + sampling + the device env's step as one launch, the previous
        step's bookkeeping riding in the normalisation launch: xrl_post_norm -- four launches per vector step instead of seven)?  A
        device CartPole / Pendulum / MountainCar / Acrobot provider, an actor-critic whose heads are single layers on two disjoint
        blocks of one hidden level (K <= 128, a multiple of 32; A <= 8), no callback between the steps.  config.use_device_act_tail,
        default False: bit-identical to the launches it replaces and measured no faster (tools/probe_act_tail.py: 10.4 us without the
        env's step against 7.1 + 3.1 for the heads' product and xrl_policy_sample -- two waves per workgroup walk staging, products,
        sampling and the simulator one latency after the other; profiles/r06_l_act_tail.json).  Returns None or (env_kind, actor head
        layer, critic head layer)."""
        if self._per_step():
            return None
        if not bool(_get(self.config, "use_device_act_tail", False)):
            return None
        if not hasattr(self, "_dtail"):
            self._dtail = None
            env, m = self.envs, self.model
            kind = getattr(env, "kind", 0) if hasattr(env, "_kw") and type(env).__name__ != "DeviceCartPoleVecEnv" else 0
            if type(env).__name__ == "DeviceCartPoleVecEnv":
                kind = 4
            plan = getattr(m, "plan", None)
            ok = type(self)._enqueue_step in (PPO_Agent._enqueue_step,) and \
                kind in (1, 2, 3, 4) and plan is not None and len(plan.stages) >= 2 and len(plan.stages[-1]) == 2
            if ok:
                La, Lc = plan.stages[-1]
                if Lc.N != 1:
                    La, Lc = Lc, La
                A = m.action_dim
                lvl = len(plan.widths) - 1
                ok = La.N == A and Lc.N == 1 and A <= 8 and La.K == Lc.K and La.K % 32 == 0 and 32 <= La.K <= 128 and \
                    La.in_level == Lc.in_level == lvl - 1 and La.out_level == Lc.out_level == lvl and La.out_off == 0 and Lc.out_off == A and \
                    Lc.act is None and plan.widths[lvl] == A + 1 and \
                    (La.in_off + La.K <= Lc.in_off or Lc.in_off + Lc.K <= La.in_off) and max(La.in_off, Lc.in_off) + La.K <= 256
                if ok:
                    self._dtail = (kind, La, Lc)
        return self._dtail

    def _post_norm_ok(self):
        """May step t's bookkeeping ride in step t + 1's statistics / normalisation launch (xrl_post_norm: both are single workgroups
        of reductions over the env axis, 10.2 us together against 7.0 + 4.6 us as two launches)?  The general path (no frames, no wide
        acting launch), no callback between the steps; config.use_post_norm: False keeps the two launches."""
        return not self.frames and not self._per_step() and bool(_get(self.config, "use_post_norm", True)) and \
            type(self)._enqueue_step is PPO_Agent._enqueue_step

    def _enqueue_step_device_tail(self, t, tail):
        """_enqueue_step with step t - 1's bookkeeping in this step's first launch (xrl_post_norm; the last step's: in
        _enqueue_rollout_tail) and, with `tail`, heads + sample + env step as one launch (xrl_act_tail).  Same numbers as the seven
        launches (tests/test_gpu_agent.py)."""
        env, mem, n, D, A = self.envs, self.memory, self.n_envs, self.obs_dim, self.model.action_dim
        f, m = mem.soa.fields, self.model
        plan, P = getattr(m, "plan", None), m.params
        gaussian = m.dist == "gaussian"
        stats = (self.obs_mean, self.obs_var, self.obs_count)
        rms = dict(x=env.buf_obs, mean=self.obs_mean, var=self.obs_var, count=self.obs_count, out0=self.X, out1=f["observations"][t],
                   n=n, D=D, ld_x=D, ld0=D, ld1=D, update=int(self.use_obsnorm), normalize=int(self.use_obsnorm),
                   range=float(self.obsnorm_range))
        if t > 0:
            ops.post_norm(self._post_args(t - 1, stats, self.X[n:]), rms)
        else:
            ops.obs_normalize(**rms)
        if tail is None:
            self._act_sample(2 * n, log_std=P.ptr("actor.log_std") if gaussian else None,
                              noise=None if self.action_noise is None else self.action_noise[t],
                              act_out=f["actions"][t], val_out=f["values"][t], logp_out=f["aux_old_logp"][t],
                              env_action=None if gaussian else env.action, env_action_f=env.action if gaussian else None,
                              bootv_prev=f["bootv"][t - 1] if t > 0 else None, n=n, A=A, ld=A + 1, gaussian=int(gaussian),
                              seed=self.seed, step=t, step_dev=self.step_counter)
            if hasattr(env, "advance"):
                env.step_device(offset=t)
            else:
                env.step_device()
            return
        kind, La, Lc = tail
        plan.forward(self.X, D, 2 * n, stages=plan.stages[:-1])
        lvl = La.in_level
        ekw = env._kw()
        ops.act_tail(sample=dict(heads=None, log_std=P.ptr("actor.log_std") if gaussian else None,
                                 noise=None if self.action_noise is None else self.action_noise[t],
                                 act_out=f["actions"][t], val_out=f["values"][t], logp_out=f["aux_old_logp"][t],
                                 env_action=None if gaussian else env.action, env_action_f=env.action if gaussian else None,
                                 bootv_prev=f["bootv"][t - 1] if t > 0 else None, n=n, A=A, ld=A + 1, gaussian=int(gaussian),
                                 seed=self.seed, step=t, step_dev=self.step_counter),
                     env_kind=kind, classic=ekw if kind != 4 else None, cartpole=ekw if kind == 4 else None,
                     hb=plan.acts[lvl], ldh=plan.widths[lvl], K=La.K, a_off=La.in_off, c_off=Lc.in_off,
                     w_actor=P.ptr(La.w_name), b_actor=P.ptr(La.b_name), w_critic=P.ptr(Lc.w_name), b_critic=P.ptr(Lc.b_name),
                     ldw_a=La.K, ldw_c=Lc.K, heads=plan.acts[La.out_level], boot_rows=1, boot_actor=0, act_actor=ops.ACT[La.act])

    def _enqueue_step(self, t):
        if self.frames:
            return self._enqueue_step_frames(t)
        env, mem, n, D, A = self.envs, self.memory, self.n_envs, self.obs_dim, self.model.action_dim
        f = mem.soa.fields
        gaussian = self.model.dist == "gaussian"
        wide = self._wide_acting()
        if wide is None and self._post_norm_ok():
            return self._enqueue_step_device_tail(t, self._device_tail())
        fold = wide is not None and self._wstats is not None
        stats = (self.obs_mean, self.obs_var, self.obs_count)
        if not fold:
            # obs_rms.update(obs); obs = _process_observation(obs); memory.observations[t] = obs   (ppo_agent.py:114-115,128)
            ops.obs_normalize(x=env.buf_obs, mean=self.obs_mean, var=self.obs_var, count=self.obs_count, out0=self.X,
                              out1=f["observations"][t], n=n, D=D, ld_x=D, ld0=D, ld1=D, update=int(self.use_obsnorm),
                              normalize=int(self.use_obsnorm), range=float(self.obsnorm_range))
        assert wide is None or self.action_noise is None, "supplied action noise needs the layered acting step (use_fused_acting: False)"
        if wide is not None:
            # forward of both branches + sample + log-prob + values in ONE launch (csrc/ppo_wide.hip: wide_act_kernel); with
            # `fold` the running statistics + normalisation as well (two statistics sets alternate: the workgroups of a
            # launch read one while the other is written; horizon_size is even, so set 0 is current between rollouts)
            kw = {}
            if fold:
                stats = self._wstats[(t + 1) & 1]
                kw = dict(raw=env.buf_obs, stats_in=self._wstats[t & 1], stats_out=stats, obs_slot=f["observations"][t],
                          update=int(self.use_obsnorm), normalize=int(self.use_obsnorm), obs_range=float(self.obsnorm_range))
                if self._wpost and t > 0:                   # the previous step's bookkeeping + its next observations, raw
                    kw.update(next_raw=env.next_obs, post=self._post_args(t - 1, self._wstats[t & 1]))
            wide.act(self.X, n, self.seed, t, self.step_counter, act_out=f["actions"][t], env_action_f=env.action,
                     logp_out=f["aux_old_logp"][t], val_out=f["values"][t], bootv_prev=f["bootv"][t - 1] if t > 0 else None, **kw)
        else:
            # actions / log-probs / values of rows [0,n) -> buffer slot t; value of rows [n,2n) -> bootv[t-1]
            self._act_sample(2 * n, log_std=self.model.params.ptr("actor.log_std") if gaussian else None,
                              noise=None if self.action_noise is None else self.action_noise[t],
                              act_out=f["actions"][t], val_out=f["values"][t], logp_out=f["aux_old_logp"][t],
                              env_action=None if gaussian else env.action, env_action_f=env.action if gaussian else None,
                              bootv_prev=f["bootv"][t - 1] if t > 0 else None, n=n, A=A, ld=A + 1, gaussian=int(gaussian),
                              seed=self.seed, step=t, step_dev=self.step_counter)
        if hasattr(env, "advance"):
            env.step_device(offset=t)                           # static step index: the env's counter ticks once per rollout
        else:
            env.step_device()
        if fold and self._wpost:
            return                                          # (rides in the next acting launch / the bootstrap launch)
        ops.rollout_poststep(**self._post_args(t, stats, self.X[n:]))

    def _post_args(self, t, stats, next_obs_norm=None):
        """Keyword arguments of xrl_rollout_poststep for vector step t (stats: the observation statistics after that step's
        update; next_obs_norm None: the consumer normalises the next observations itself)."""
        env, f, n, D = self.envs, self.memory.soa.fields, self.n_envs, self.obs_dim
        return dict(reward=env.reward, terminated=env.terminated, truncated=env.truncated, next_obs=env.next_obs,
                    obs_mean=stats[0], obs_var=stats[1], next_obs_norm=next_obs_norm, rew_out=f["rewards"][t],
                    term_out=f["terminals"][t], seg_out=f["seg"][t], ret_track=self.returns,
                    ret_mean=self.ret_mean, ret_var=self.ret_var, ret_count=self.ret_count, n=n, D=D, ld_next=D,
                    use_obsnorm=int(self.use_obsnorm), use_rewnorm=int(self.use_rewnorm),
                    last_step=int(t == self.horizon_size - 1), obs_range=float(self.obsnorm_range),
                    rew_range=float(self.rewnorm_range), gamma=float(self.gamma))

    def _enqueue_rollout_fused(self, kernel_only=False):
        """The whole rollout on the device + GAE: same numbers as _enqueue_rollout (the layered path).  kernel_only: just the
        rollout kernel(s) (bench.py times them in isolation)."""
        T, n, D, A = self.horizon_size, self.n_envs, self.obs_dim, self.model.action_dim
        env, f, pp = self.envs, self.memory.soa.fields, self.pp
        cpr = self._actor_rollout()
        if cpr is not None:
            # the 4-128-{128-2,128-1} class (csrc/rollout_actor.hip): only the actor is on the step chain; ONE launch for all
            # T steps (resident workgroups, the step boundary is one tagged message per workgroup), or -- per-step callbacks,
            # or after a launch reported a time-out -- one launch per vector step of the same kernel (bit-identical); values
            # and bootstrap values of the whole rollout follow as one batched launch
            if self._persistent_ok():
                cpr.run(0, T, flags=int(bool(_get(self.config, "persistent_coherent_exchange", False))),
                        dbg=getattr(self, "rollout_dbg", None))
                cpr.values(0, T)
            else:
                hooks = self._per_step() and not kernel_only
                for t in range(T):
                    cpr.run(t, 1)
                    if hooks:
                        cpr.values(t, 1)
                        self._step_hooks(t)
                if not hooks:
                    cpr.values(0, T)
            if kernel_only:
                return
            ops.counter_add(self.step_counter, T)
            if getattr(env, "tape", None) is not None:
                ops.counter_add(env.tape_pos, T)                # the next rollout reads the following stretch of the tape
            ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                         self.gamma, self.gae_lam, self.memory.use_gae)
            return
        assert getattr(env, "tape", None) is None, "a tape provider needs the actor rollout kernel (4-128-{128-2,128-1}, n_envs <= 256)"
        # any other shape / more envs: T launches of the any-shape step kernel + one bootstrap-only launch
        if not kernel_only:
            ops.pack_rollout_cache(self.model.plan, self.model.params.flat, self.cache_image, self.frag_image)   # params changed
        plan = self.model.plan
        mids = [L for st in plan.stages[1:-1] for L in st]
        heads = plan.stages[-1]
        split_ok = bool(_get(self.config, "use_role_split", True)) and len(mids) == 1 and len(heads) == 2 and \
            mids[0].N == plan.widths[mids[0].out_level] and heads[1].in_off % 32 == 0 and heads[1].in_off > 0
        common = dict(params=self.model.params.flat, cache_image=self.cache_image, frag_image=self.frag_image,
                      role_split=int(split_ok), split_col=int(heads[1].in_off) if split_ok else 0, ret_track=self.returns, cp_state=env.state, cp_steps=env.steps,
                      cp_episodes=env.episodes, cp_score=env.ep_score, cp_stats=env.stats, n=n, D=D, A=A, gaussian=0,
                      max_steps=int(env.max_episode_steps), use_obsnorm=int(self.use_obsnorm),
                      use_rewnorm=int(self.use_rewnorm), obs_range=float(self.obsnorm_range),
                      rew_range=float(self.rewnorm_range), gamma=float(self.gamma), seed=self.seed, env_seed=env.seed,
                      step_dev=self.step_counter)
        for t in range(T):
            i, o = t & 1, (t + 1) & 1
            ops.rollout_step_cartpole(
                self.model.plan, obs_raw_in=pp["obs_raw"][i], obs_raw_out=pp["obs_raw"][o], xnext_in=pp["xnext"][i],
                xnext_out=pp["xnext"][o], obs_stats_in=pp["obs_stats"][i], obs_stats_out=pp["obs_stats"][o],
                obs_count_in=pp["obs_count"][i], obs_count_out=pp["obs_count"][o], ret_stats_in=pp["ret_stats"][i],
                ret_stats_out=pp["ret_stats"][o], ret_count_in=pp["ret_count"][i], ret_count_out=pp["ret_count"][o],
                ended_in=pp["ended"][i], ended_out=pp["ended"][o], ret_final_in=pp["ret_final"][i],
                ret_final_out=pp["ret_final"][o], obs_slot=f["observations"][t], act_slot=f["actions"][t],
                val_slot=f["values"][t], logp_slot=f["aux_old_logp"][t], rew_slot=f["rewards"][t],
                term_slot=f["terminals"][t], seg_slot=f["seg"][t], bootv_prev=f["bootv"][t - 1] if t > 0 else None,
                last_step=int(t == T - 1), boot_only=0, step=t, **common)
            if not kernel_only:
                self._step_hooks(t)
        ops.rollout_step_cartpole(self.model.plan, xnext_in=pp["xnext"][T & 1], bootv_prev=f["bootv"][T - 1], boot_only=1,
                                  last_step=0, step=0, **common)
        if kernel_only:
            return
        ops.counter_add(self.step_counter, T)
        ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                     self.gamma, self.gae_lam, self.memory.use_gae)

    def _per_step(self):
        """config.per_step_callbacks with a callback that overrides on_train_step / on_train_step_end: the rollout runs as eager
        per-step launches (no whole-rollout launch, no rollout graph) and the hooks fire after every vector step."""
        if not hasattr(self, "_per_step_cb"):
            self._per_step_cb = bool(_get(self.config, "per_step_callbacks", False)) and \
                (self._has_cb("on_train_step") or self._has_cb("on_train_step_end"))
        return self._per_step_cb

    def _step_hooks(self, t):
        """ppo_agent.py:123-126,179-180 for vector step t of the running rollout: the buffer's slot t as device tensors."""
        if not self._per_step():
            return
        f, n = self.memory.soa.fields, self.n_envs
        step = self.current_step + t * n
        self._cb("on_train_step", step, envs=self.envs, policy=self.model, obs=f["observations"][t], acts=f["actions"][t],
                 vals=f["values"][t], rewards=f["rewards"][t], terminals=f["terminals"][t], aux_info={"old_logp": f["aux_old_logp"][t]},
                 next_obs=getattr(self.envs, "next_obs", None), truncations=getattr(self.envs, "truncated", None), infos=None,
                 train_steps=getattr(self, "_train_steps", None))
        self._cb("on_train_step_end", step + n, envs=self.envs, policy=self.model, train_steps=getattr(self, "_train_steps", None),
                 train_info=getattr(self, "_last_info", {}))

    def _actor_rollout(self):
        """ops.CartPoleRollout of this agent when the network is the 4-128-{128-2,128-1} class and n_envs <= 256
        (config.use_actor_rollout: False keeps the any-shape per-step kernel), else None.  State lives in slot 0 of the
        ping-pong tensors (where the any-shape path leaves it after an even number of steps)."""
        if not hasattr(self, "_cpr"):
            self._cpr = None
            if bool(_get(self.config, "use_actor_rollout", True)) and ops.CartPoleRollout.eligible(self.model.plan, self.n_envs):
                env, f, pp, n, T, dev = self.envs, self.memory.soa.fields, self.pp, self.n_envs, self.horizon_size, self.device
                n4 = (n + 3) // 4 * 4
                self.persist_xchg = torch.zeros(2048, dtype=torch.int32, device=dev)
                # the status words ride in the learner's read-back block: the host sees them at the one sync of every update
                status = self.learner.status_words
                status.zero_()
                self.persist_status = status if bool(_get(self.config, "use_persistent_rollout", True)) and not self._per_step() else None
                self._cpr = ops.CartPoleRollout(
                    self.model.plan, T, params=self.model.params.flat, n=n, max_steps=int(env.max_episode_steps),
                    use_obsnorm=int(self.use_obsnorm), use_rewnorm=int(self.use_rewnorm), obs_range=float(self.obsnorm_range),
                    rew_range=float(self.rewnorm_range), gamma=float(self.gamma), seed=self.seed, env_seed=env.seed, step=0,
                    step_dev=self.step_counter, obs_raw=pp["obs_raw"][0], obs_stats=pp["obs_stats"][0], obs_count=pp["obs_count"][0],
                    ret_stats=pp["ret_stats"][0], ret_count=pp["ret_count"][0], ret_track=self.returns, cp_state=env.state,
                    cp_steps=env.steps, cp_episodes=env.episodes, cp_score=env.ep_score, cp_stats=env.stats,
                    f_obs=f["observations"], f_act=f["actions"], f_logp=f["aux_old_logp"], f_rew=f["rewards"], f_term=f["terminals"],
                    f_seg=f["seg"], f_val=f["values"], bootv=f["bootv"], xnext=torch.zeros(T, n, 4, device=dev),
                    ended=torch.zeros(T, n4, dtype=torch.uint8, device=dev), ret_final=torch.zeros(T, n4, device=dev),
                    xchg=self.persist_xchg, status=status)
                tape = getattr(env, "tape", None)
                if tape is not None:
                    # a recorded run as the provider (envs/recorded.py: TapeCartPoleVecEnv): the kernel reads the simulators' outputs from
                    # the tape and its sampling uniforms from the staging tensor set_action_noise fills (xrl_rollout_run_t.tape_*)
                    self.action_noise = torch.zeros(T, n, device=dev)
                    q = self._cpr.q
                    q.tape_next_obs, q.tape_reset_obs = tape["next_obs"].data_ptr(), tape["reset_obs"].data_ptr()
                    q.tape_term, q.tape_trunc = tape["term"].data_ptr(), tape["trunc"].data_ptr()
                    q.tape_pos, q.tape_rows, q.tape_u = env.tape_pos.data_ptr(), int(env.n_steps), self.action_noise.data_ptr()
        elif self._cpr is not None and not ops.fast_kernels_enabled():
            return None
        return self._cpr

    def _persistent_ok(self):
        """Whole-rollout launch of the actor kernel: unless switched off (config.use_persistent_rollout, or after a launch
        reported a time-out) or per-step callbacks want the host between the steps."""
        return bool(_get(self.config, "use_persistent_rollout", True)) and not self._per_step() and \
            getattr(self, "persist_status", None) is not None

    def _wide_rollout(self):
        """ops.WideRollout of this agent -- the whole rollout as one launch with only the actor on the step chain
        (csrc/rollout_wide.hip) -- when the network is the two-branch Gaussian class, the env the device-resident continuous-control
        provider and n_envs <= 256 (config.use_wide_rollout: False keeps the launches per vector step), else None."""
        if not hasattr(self, "_wr"):
            self._wr = None
            from ..envs.synthetic import SyntheticMujocoVecEnv
            env, n, T, dev = self.envs, self.n_envs, self.horizon_size, self.device
            ok = bool(_get(self.config, "use_wide_rollout", True)) and type(self)._enqueue_step is PPO_Agent._enqueue_step and \
                (type(env) is SyntheticMujocoVecEnv or getattr(env, "is_control_tape", False)) and self.model.dist == "gaussian" and \
                ops.WideRollout.eligible(self.model, n) and \
                tuple(env.buf_obs.shape) == (n, self.obs_dim)
            if ok:
                f, D = self.memory.soa.fields, self.obs_dim
                n4 = (n + 3) // 4 * 4
                self._wr_xnext = torch.zeros(T * n, D, device=dev)
                self._wr_xchg = torch.zeros(ops.WideRollout.xchg_words(), dtype=torch.int32, device=dev)
                # the status words ride in the learner's read-back block: the host sees them at the one sync of every update
                self._wr_status = self.learner.status_words
                self._wr_status.zero_()
                self.model.plan.ensure(T * n)                     # (the batched values pass; before any capture)
                self._wr = ops.WideRollout(
                    self.model, T, params=self.model.params.flat, n=n, max_steps=int(env.max_episode_steps),
                    use_obsnorm=int(self.use_obsnorm), use_rewnorm=int(self.use_rewnorm), obs_range=float(self.obsnorm_range),
                    rew_range=float(self.rewnorm_range), gamma=float(self.gamma), seed=self.seed, env_seed=env.seed, step=0, env_step=0,
                    step_dev=self.step_counter, env_step_dev=env.step_counter, obs_raw=env.buf_obs, obs_mean=self.obs_mean,
                    obs_var=self.obs_var, obs_count=self.obs_count, ret_mean=self.ret_mean, ret_var=self.ret_var, ret_count=self.ret_count,
                    ret_track=self.returns, env_state=env.state, env_steps=env.steps, env_score=env.ep_score, env_stats=env.stats,
                    Amat=env.A, Bmat=env.B, f_obs=f["observations"], f_act=f["actions"], f_logp=f["aux_old_logp"], f_rew=f["rewards"],
                    f_term=f["terminals"], f_seg=f["seg"], xnext=self._wr_xnext, ended=torch.zeros(T, n4, dtype=torch.uint8, device=dev),
                    ret_final=torch.zeros(T, n4, device=dev), raw_rew=torch.zeros(T, n4, device=dev), xchg=self._wr_xchg,
                    status=self._wr_status)
                tape = getattr(env, "tape", None)
                if tape is not None:
                    # a recorded run as the provider (envs/recorded.py: TapeControlVecEnv): the simulators' outputs from the tape, the action
                    # draws' normals from the staging tensor set_action_noise fills (xrl_rollout_wide_t.tape_*)
                    self.action_noise = torch.zeros(T, n, self.model.action_dim, device=dev)
                    q = self._wr.q
                    q.tape_next_obs, q.tape_reset_obs, q.tape_rew = (tape[k].data_ptr() for k in ("next_obs", "reset_obs", "rew"))
                    q.tape_term, q.tape_trunc = tape["term"].data_ptr(), tape["trunc"].data_ptr()
                    q.tape_pos, q.tape_rows, q.tape_z = env.tape_pos.data_ptr(), int(env.n_steps), self.action_noise.data_ptr()
        return self._wr

    def _enqueue_rollout_wide(self, wr):
        """The whole rollout of the two-branch Gaussian class: one launch for the T vector steps (actor chain, dynamics, statistics,
        records), then values and bootstrap values of every stored row as two batched forward passes (the layered GEMM plan at
        T x n rows) -- same numbers as the launches per vector step up to fp32 summation order."""
        T, n, A, f = self.horizon_size, self.n_envs, self.model.action_dim, self.memory.soa.fields
        M = T * n
        if self._per_step():
            for t in range(T):
                wr.run(t, 1)
        else:
            wr.run(0, T)
        heads = self.model.forward_values(f["observations"].view(M, -1), M)
        ops.copy_column(heads, A + 1, A, f["values"], M)
        heads = self.model.forward_values(self._wr_xnext, M)
        ops.copy_column(heads, A + 1, A, f["bootv"], M)
        self.envs.advance(T)
        ops.counter_add(self.step_counter, T)
        ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                     self.gamma, self.gae_lam, self.memory.use_gae)

    def _enqueue_rollout(self):
        if self.use_fused_rollout:
            return self._enqueue_rollout_fused()
        wr = self._wide_rollout()
        if wr is not None:
            return self._enqueue_rollout_wide(wr)
        T, n, A = self.horizon_size, self.n_envs, self.model.action_dim
        # frame stacks: the parameters do not change inside a rollout, so the convolution weight images are built ONCE (not by every
        # vector step's pass: 5 us each) and the dense layer reads the last convolution's output in place through a column-permuted
        # copy of its weights (no flatten launch: 10.8 us per step); config.use_fast_frame_acting
        conv = getattr(self.model, "conv", None)
        fast = self.frames and conv is not None and conv.implicit and hasattr(self.model, "refresh_acting_params") and \
            bool(_get(self.config, "use_fast_frame_acting", True))
        if fast:
            self.model.refresh_acting_params()
            conv.invalidate()
            conv.pack_images([(None, False)])
            conv.mark_live(self.model.params.flat)
            self._acting_fast = True
        try:
            if not self.frames and self._trunk_forward() is not None:    # the acting pass's weight planes from the parameters of the moment
                ops.pack_mid_frags16(self.model.plan, self.model.params.flat, self._trunk_forward())
            for t in range(T):
                self._enqueue_step(t)
                self._step_hooks(t)
            if hasattr(self.envs, "advance"):
                self.envs.advance(T)
            self._enqueue_rollout_tail(T, n, A)
        finally:
            if fast:
                self._acting_fast = False
                conv.invalidate()

    def _enqueue_rollout_tail(self, T, n, A):
        # buffer full: vals = get_terminated_values(next_obs) for every env (ppo_agent.py:129-135)
        wide = self._wide_acting()
        if wide is not None:
            kw = {}
            if self._wstats is not None and self._wpost:    # step T - 1's bookkeeping and raw next observations (T is even:
                st = self._wstats[0]                        #  set 0 holds the current statistics)
                kw = dict(next_raw=self.envs.next_obs, post=self._post_args(T - 1, st), stats_in=st,
                          normalize=int(self.use_obsnorm), obs_range=float(self.obsnorm_range))
            wide.act(self.X, n, self.seed, 0, None, bootv_prev=self.memory.soa.fields["bootv"][T - 1], **kw)
        elif self.frames and self._frame_tail():
            # the last step's bookkeeping rides in the bootstrap pass's tail launch (xrl_ppo_act_tail with act_out = None: values only)
            self.model.act_tail(self._policy_frames(), 2 * n, n, post=self._post_args(T - 1, (self.obs_mean, self.obs_var, self.obs_count), None),
                                act_out=None, val_out=None, logp_out=None, env_action=None,
                                bootv_prev=self.memory.soa.fields["bootv"][T - 1], seed=self.seed, step=0, step_dev=None)
        else:
            if self._wide_acting() is None and self._post_norm_ok():   # the last step's bookkeeping (rode in the next step's launch so far)
                ops.rollout_poststep(**self._post_args(T - 1, (self.obs_mean, self.obs_var, self.obs_count), self.X[n:]))
            if self.frames:
                heads = self.model.forward(self._policy_frames(), 2 * n, keep=False, acting=self._acting_fast)
                ops.policy_sample(heads=heads, act_out=None, val_out=None, logp_out=None,
                                  bootv_prev=self.memory.soa.fields["bootv"][T - 1], n=n, A=A, ld=A + 1,
                                  gaussian=0, seed=self.seed, step=0, step_dev=None)
            else:
                self._act_sample(2 * n, act_out=None, val_out=None, logp_out=None, bootv_prev=self.memory.soa.fields["bootv"][T - 1],
                                 n=n, A=A, ld=A + 1, gaussian=0, seed=self.seed, step=0, step_dev=None)
        ops.counter_add(self.step_counter, T)
        f = self.memory.soa.fields
        ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                     self.gamma, self.gae_lam, self.memory.use_gae)

    def _enqueue_update(self):
        """train_epochs (core/on_policy.py:182-205): n_epochs x n_minibatch minibatches taken from self.idx."""
        mem, lr = self.memory, self.learner
        if self.rem:
            return self._enqueue_update_ragged()
        nb, bs = self.idx.shape
        f = mem.soa
        fused = lr.fused_eligible(mem)
        if fused:
            lr.prepare_fused(mem, bs)
        else:
            lr.prepare_buffer_update(mem, bs)
        if not getattr(self, "_fixed_idx", False):
            self._new_indices()
        if fused:
            lr.refresh_fused_params(mem, self.idx)
        if mem.use_advnorm:
            ops.adv_stats(f.fields["advantages"], self.idx.view(-1), bs, nb, self.n_envs, self.horizon_size, lr.stats)
        if fused:
            # (the optimiser step of minibatch k rides in the launch of minibatch k + 1 where the learner can chain them)
            for k in range(nb):
                lr.enqueue_minibatch_fused(mem, self.idx[k], lr.stats[k] if mem.use_advnorm else None, defer=True)
            lr.finish_pending()
            return
        for k in range(nb):
            lr.enqueue_minibatch_from_buffer(mem, self.idx[k], lr.stats[k] if mem.use_advnorm else None)

    def _enqueue_update_ragged(self):
        """buffer_size % n_minibatch != 0: per epoch n_minibatch full minibatches and one short one (layered path)."""
        mem, lr, bs = self.memory, self.learner, self.batch_size
        lr.prepare_buffer_update(mem, bs)
        if not getattr(self, "_fixed_idx", False):
            ops.random_permutation(self.idx_full, self.n_epochs, self.buffer_size, self.buffer_size, self.seed, 0, self.perm_counter)
            ops.counter_add(self.perm_counter, 1)
        adv, k = mem.soa.fields["advantages"], 0
        for e in range(self.n_epochs):
            for start in range(0, self.buffer_size, bs):
                idx = self.idx_full[e, start:start + bs]
                st = None
                if mem.use_advnorm:
                    st = lr.stats[k]
                    ops.adv_stats(adv, idx, idx.numel(), 1, self.n_envs, self.horizon_size, st)
                lr.enqueue_minibatch_from_buffer(mem, idx, st)
                k += 1

    def _new_indices(self):
        """np.random.shuffle of arange(buffer_size) per epoch (on_policy.py:194-204), generated on the device."""
        # one launch, part of the captured update graph: a keyed bijection per epoch (xrl_random_permutation), keyed by a
        # device counter that ticks once per update phase
        ops.random_permutation(self.idx, self.n_epochs, self.buffer_size, self.n_minibatch * self.batch_size, self.seed,
                               0, self.perm_counter)
        ops.counter_add(self.perm_counter, 1)

    # -- public API -----------------------------------------------------------------------------------------------
    def set_action_noise(self, noise):
        """Parity / replay hook: the next rollout's action draws come from `noise` ([horizon_size, n] uniforms for a categorical
        policy -- the action is the inverse CDF of the softmax at that uniform --, [horizon_size, n, A] standard normals for a
        Gaussian one) instead of the Philox stream.  The values are copied into one staging tensor, so a captured rollout graph
        replays on whatever the caller staged last."""
        x = torch.as_tensor(np.asarray(noise, np.float32), device=self.device)
        if getattr(self.envs, "tape", None) is not None and (self.use_fused_rollout or getattr(self.envs, "is_control_tape", False)):
            # the one-launch rollouts over a tape read their draws from the staging tensor (xrl_rollout_run_t.tape_u / xrl_rollout_wide_t.tape_z)
            assert (self._actor_rollout() if self.use_fused_rollout else self._wide_rollout()) is not None, \
                "a tape provider needs the one-launch rollout kernel of the network's class"
            self.action_noise.copy_(x.reshape(self.action_noise.shape))
            return
        assert not self.use_fused_rollout and self._wide_rollout() is None, \
            "supplied action noise needs the layered rollout (use_fused_rollout / use_wide_rollout: False)"
        if self.action_noise is None:
            self.action_noise = torch.zeros_like(x).contiguous()
            self._rollout_graph = None                      # (a graph captured before drew from the Philox stream)
        self.action_noise.copy_(x.reshape(self.action_noise.shape))

    def set_indices(self, idx):
        """Parity hook: use the caller's minibatch indices (e.g. the ones NumPy produced for the reference); with a
        remainder ([n_epochs, buffer_size]: every epoch's whole permutation)."""
        if self.rem:
            self.idx_full.copy_(torch.as_tensor(np.asarray(idx)).reshape(self.idx_full.shape))
        else:
            self.idx.copy_(torch.as_tensor(np.asarray(idx)).reshape(self.idx.shape))
        if not getattr(self, "_fixed_idx", False):
            self._update_graph = None                     # a captured graph would regenerate the indices
            self._mb_graphs = None
        self._fixed_idx = True

    # -- whole-rollout launch: its failure flags (csrc/rollout_actor.hip) are part of the contract -----------------------
    @staticmethod
    def _persist_status_ok(st):
        """status[0] != 0: a barrier timed out (results invalid).  Workgroups on more than one XCD are NOT a failure: the
        kernel notices in every launch and exchanges through device-scope stores then (status[3] counts those launches)."""
        return st[0] == 0

    def _wide_acting(self):
        """The learner's PpoWideState when the acting step of the layered rollout can run as one launch (the two-branch
        Gaussian class of csrc/ppo_wide.hip, PPO-clip learner; config.use_fused_acting: False switches it off), else None.
        Its fragment-ordered copy of the middle layers is the one the update phase keeps current."""
        if not hasattr(self, "_wact"):
            lr = self.learner
            ok = type(self)._enqueue_step is PPO_Agent._enqueue_step and bool(_get(self.config, "use_fused_acting", True)) and \
                hasattr(lr, "wide_eligible") and lr.wide_eligible()
            if ok:
                lr._wide_prepare(self.batch_size)
                lr._wide.prepare_act(self.n_envs)
            self._wact = lr._wide if ok else None
            # running statistics + normalisation inside the acting launch: needs a second statistics set, an even horizon
            # (set 0 = obs_mean / obs_var / obs_count is then current whenever the host looks) and few enough rows
            self._wstats = None
            D = self.obs_dim
            if ok and bool(_get(self.config, "use_fused_obsnorm", True)) and self.horizon_size % 2 == 0 and self.n_envs % 32 == 0 and \
                    self.n_envs <= 4 * (1024 // D) and tuple(self.envs.buf_obs.shape) == (self.n_envs, D):
                self._wstats = [(self.obs_mean, self.obs_var, self.obs_count),
                                (torch.zeros_like(self.obs_mean), torch.ones_like(self.obs_var), torch.zeros_like(self.obs_count))]
            # ... and the previous step's bookkeeping as one more workgroup of the acting launch (rows [n, 2n) then come as raw
            # next observations; tiles must not straddle row n)
            self._wpost = self._wstats is not None and bool(_get(self.config, "use_fused_poststep", True))
        return self._wact

    def _rollout_state_tensors(self):
        """Everything a rollout launch mutates besides the buffer slots it overwrites (simulator, statistics, counters)."""
        env, t = self.envs, [self.returns, self.step_counter]
        t += [getattr(env, k) for k in ("state", "steps", "episodes", "ep_score", "stats", "buf_obs")]
        if getattr(env, "tape", None) is not None:
            t.append(env.tape_pos)
        return t + list(self.pp.values())

    def _ws_sig(self):
        """Changes whenever a workspace a captured graph may point into was reallocated (Plan.ensure grew, a convolution
        workspace was replaced by a larger one -- e.g. get_actions on more rows than the loops use)."""
        return (self.model.plan.cap, getattr(getattr(self.model, "conv", None), "ws_gen", 0))

    def _launch_rollout(self):
        if not self.use_fused_rollout:
            self._wide_acting()                                   # (allocates on first use: never inside a capture)
            self._wide_rollout()
            if not self.frames:
                self._trunk_forward()                             # (may prepare the learner's fused state: never inside a capture)
        safe = getattr(self.envs, "graph_safe", True) or (getattr(self.envs, "graph_safe_even", False) and self.horizon_size % 2 == 0)
        if self.use_graph and safe and not self._per_step():
            if self._rollout_graph is not None and self._ws_sig() != self._rollout_cap:
                self._rollout_graph = None        # the dense workspaces grew since the capture (a larger get_actions batch):
            if self._rollout_graph is None:       # the old graph holds freed pointers -- capture again
                if self.frames:                                       # (workspaces of the convolution stack: allocated outside the capture)
                    self.model.forward(self._policy_frames(), 2 * self.n_envs, keep=False)
                    if hasattr(self.model, "refresh_acting_params"):  # (... and the acting copy of the dense parameters)
                        self.model.refresh_acting_params()
                        self.model.forward(self._policy_frames(), 2 * self.n_envs, keep=False, acting=True)
                torch.cuda.synchronize()
                g = ops.Graph()
                with g:
                    self._enqueue_rollout()
                self._rollout_graph = g
                self._rollout_cap = self._ws_sig()
            self._rollout_graph.launch()
        else:
            self._enqueue_rollout()

    def rollout(self):
        if not self._started:
            self.envs.reset()
            if self.use_fused_rollout:
                self.pp["obs_raw"][0].copy_(self.envs.buf_obs)
            self._started = True
        first = self.use_fused_rollout and not getattr(self, "_persist_checked", False)
        saved = [x.clone() for x in self._rollout_state_tensors()] if first else None
        wide_first = not self.use_fused_rollout and not getattr(self, "_wide_checked", False) and self._wide_rollout() is not None
        if wide_first:
            # first whole-rollout launch of the Gaussian class: read its status synchronously (a wait between the resident workgroups
            # that timed out leaves the rollout incomplete): restore, fall back to the launches per vector step for good, redo
            self._wide_checked = True
            env = self.envs
            wstate = [self.returns, self.step_counter, env.step_counter, env.state, env.steps, env.ep_score, env.stats, env.buf_obs,
                      self.obs_mean, self.obs_var, self.obs_count, self.ret_mean, self.ret_var, self.ret_count]
            if getattr(env, "tape", None) is not None:
                wstate.append(env.tape_pos)
            wsaved = [x.clone() for x in wstate]
            self._launch_rollout()
            torch.cuda.synchronize()
            if self._wr_status.tolist()[0] != 0:
                import warnings
                warnings.warn(f"xuance_amd: whole-rollout launch of the Gaussian class unusable on this device (status "
                              f"{self._wr_status.tolist()}); falling back to the launches per vector step")
                for x, s0 in zip(wstate, wsaved):
                    x.copy_(s0)
                self._wr, self._rollout_graph = None, None
                self.config.use_wide_rollout = False
                self._launch_rollout()
            self.current_step += self.n_envs * self.horizon_size
            return
        self._launch_rollout()
        if first:
            # First rollout of this agent: if it went through the whole-rollout launch, read its status synchronously.
            # A device where the surviving workgroups do not share one L2 (partitioned modes, another dispatch order)
            # or a barrier time-out shows here -- then the state is restored, the agent falls back to per-step launches
            # for good and redoes the rollout.  Later rollouts are checked at the read-back of every update phase.
            self._persist_checked = True
            if getattr(self, "persist_status", None) is not None:
                torch.cuda.synchronize()
                st = self.persist_status.tolist()
                if not self._persist_status_ok(st):
                    import warnings
                    warnings.warn(f"xuance_amd: whole-rollout launch unusable on this device (status {st}); "
                                  "falling back to one launch per vector step")
                    for x, s0 in zip(self._rollout_state_tensors(), saved):
                        x.copy_(s0)
                    self.config.use_persistent_rollout = False
                    self.persist_status.zero_()
                    self.persist_status = None
                    self._rollout_graph = None
                    self._launch_rollout()
        self.current_step += self.n_envs * self.horizon_size

    def _update_distributed(self):
        """N > 1 ranks.  The gradient all-reduce is the only thing between two minibatches that cannot simply be replayed
        from a graph of this process alone, so the update phase is cut AT the collectives and nowhere else: graph 0 =
        [indices, advantage statistics, minibatch 0 up to its reduced gradient], graph k = [optimiser step of minibatch
        k - 1 on the averaged gradient, minibatch k up to its reduced gradient], graph nb = [the last optimiser step]:
        nb + 1 graph launches and nb collectives per phase.  With `dist_graph_collective` (validated once per process by
        dist.collective_capturable) the RCCL calls are captured as well and the whole phase is ONE graph, as on one GPU."""
        assert not self.rem, "multi-GPU updates need buffer_size divisible by n_minibatch"
        from .. import dist as xdist
        mem, lr = self.memory, self.learner
        nb, bs = self.idx.shape
        fused = lr.fused_eligible(mem)
        step = lr.enqueue_minibatch_fused if fused else lr.enqueue_minibatch_from_buffer

        def head(k):
            if k == 0 and not getattr(self, "_fixed_idx", False):
                self._new_indices()
            if k == 0 and mem.use_advnorm:
                ops.adv_stats(mem.soa.fields["advantages"], self.idx.view(-1), bs, nb, self.n_envs,
                              self.horizon_size, lr.stats)
            if k == 0 and fused:
                lr.refresh_fused_params(mem, self.idx)
            if k > 0:
                lr.finish_after_allreduce()
            if k < nb:
                step(mem, self.idx[k], lr.stats[k] if mem.use_advnorm else None, finish=False)

        if getattr(self, "_mb_graphs", None) is None:
            if fused:
                lr.prepare_fused(mem, bs)
                lr.prepare_rows(self.idx.numel())
            else:
                lr.prepare_buffer_update(mem, bs)
            whole = bool(getattr(self.config, "dist_graph_collective", "auto")) and xdist.collective_capturable(self.device)
            torch.cuda.synchronize()
            self._mb_graphs = []
            if whole:
                g = ops.Graph()
                with g:
                    for k in range(nb + 1):
                        head(k)
                        if k < nb:
                            xdist.allreduce_mean_(lr.optimizer.grad)
                self._mb_graphs, self._whole_phase_graph = [g], True
            else:
                for k in range(nb + 1):
                    g = ops.Graph()
                    with g:
                        head(k)
                    self._mb_graphs.append(g)
                self._whole_phase_graph = False
        if self._whole_phase_graph:
            self._mb_graphs[0].launch()
            return
        for k in range(nb):
            self._mb_graphs[k].launch()                     # [Adam of k - 1,] gather + forward + loss + backward + slab reduction
            xdist.allreduce_mean_(lr.optimizer.grad)        # RCCL mean of the flat gradient
        self._mb_graphs[nb].launch()

    def update(self):
        lr = self.learner
        multi = lr.distributed_training and lr.world_size > 1
        if multi and not self.rem and lr.fused_eligible(self.memory) and lr.gradient_exchange() is not None:
            multi = False           # the ranks meet inside xrl_reduce_adam_exchange: the single-GPU update graph is the N-GPU one
        if multi:
            self._update_distributed()
        elif self.use_graph:
            if self._update_graph is not None and self._ws_sig() != self._update_cap:
                self._update_graph = None         # (see _launch_rollout)
            if self._update_graph is None:
                if self.rem:
                    self.learner.prepare_buffer_update(self.memory, self.batch_size)
                elif self.learner.fused_eligible(self.memory):
                    self.learner.prepare_fused(self.memory, self.batch_size)
                    self.learner.prepare_rows(self.idx.numel())
                else:
                    self.learner.prepare_buffer_update(self.memory, self.batch_size)
                torch.cuda.synchronize()
                g = ops.Graph()
                with g:
                    self._enqueue_update()
                self._update_graph = g
                self._update_cap = self._ws_sig()
            self._update_graph.launch()
        else:
            self._enqueue_update()
        self.learner.iterations += self.idx.shape[0] + (self.n_epochs if self.rem else 0)
        info = self.learner.last_info(self.rem if self.rem else self.batch_size)
        if getattr(self, "_wr", None) is not None and self.learner.last_status[0] != 0:
            raise ops.XrlError(f"xrl_rollout_wide_run: status {self.learner.last_status} (a wait between the resident workgroups timed "
                               "out): the last rollout is incomplete; restart with use_wide_rollout: False")
        if getattr(self, "persist_status", None) is not None and not self._persist_status_ok(self.learner.last_status):
            # read at the one host sync of the update phase: the rollout this update consumed was cut short
            raise ops.XrlError(f"xrl_rollout_cartpole_run: status {self.learner.last_status} (a wait between the resident "
                               "workgroups timed out): the last rollout is incomplete; restart with use_persistent_rollout: False")
        return info

    def train(self, train_steps):
        """Runs ``train_steps`` vector steps (rounded up to whole rollouts of horizon_size steps)."""
        info = {}
        n_rollouts = (train_steps + self.horizon_size - 1) // self.horizon_size
        self._train_steps = train_steps
        for _ in range(n_rollouts):
            self.rollout()
            info = self.update()
            self._last_info = info
            # ppo_agent.py:139-141; the per-step hooks fired inside the rollout when config.per_step_callbacks asks for them,
            # otherwise on_train_step_end fires once per rollout (kwargs: steps = the vector steps it covers)
            self._cb("on_train_epochs_end", self.current_step, policy=self.model, memory=self.memory, train_steps=train_steps,
                     update_info=info)
            if not self._per_step():
                self._cb("on_train_step_end", self.current_step, envs=self.envs, policy=self.model, train_steps=train_steps,
                         train_info=info, steps=self.horizon_size)
        eps, score, length = self.envs.episode_stats() if hasattr(self.envs, "episode_stats") else (0, 0.0, 0.0)
        info.update({"episodes": eps, "mean_episode_score": score, "mean_episode_length": length})
        return info

    # -- checkpoints: AgentSurface.save_model / load_model (agent.py:199-233) over these two hooks --------------------------
    def _obs_stats_tensors(self):
        """(mean, var, count) tensors holding the CURRENT observation statistics: the fused rollout keeps them in the
        ping-pong slot 0 between rollouts (horizon_size is even), the layered path in obs_mean / obs_var / obs_count."""
        if self.use_fused_rollout:
            D = self.obs_dim
            return self.pp["obs_stats"][0][:D], self.pp["obs_stats"][0][D:], self.pp["obs_count"][0]
        return self.obs_mean, self.obs_var, self.obs_count

    def _after_load(self):
        if self.use_fused_rollout and self.use_obsnorm:            # both ping-pong slots start from the same statistics
            self.pp["obs_stats"][1].copy_(self.pp["obs_stats"][0]); self.pp["obs_count"][1].copy_(self.pp["obs_count"][0])
        if getattr(self, "_wact", None) is not None:
            self._wact.pack()                                      # fragment-ordered copy of the loaded middle layers
        self._rollout_graph = None                                 # parameters' derived layouts are rebuilt on the next rollout/update
        self._update_graph = None
        self._mb_graphs = None

    # -- acting outside the training loop (core/on_policy.py:128-169, 303-400) ----------------------------------------------
    @torch.no_grad()
    def get_actions(self, observations, deterministic=False, return_dists=False, return_logpi=False):
        """OnPolicyAgent.get_actions: PROCESSED observations [m, obs_dim] (NumPy or device tensor) -> ActionOutput with
        NumPy env_actions / values (/ log_probs).  One forward through the same GEMM plan as the training loop; sampling by
        xrl_policy_sample (its Philox stream continues from the agent's step counter), the deterministic choice
        (CategoricalDistribution.deterministic_sample = argmax of the probabilities, Gaussian: the mean,
        distributions.py:150-153, 185-188) from the head outputs.  distributions: the head outputs as a dict when asked."""
        X = torch.as_tensor(np.asarray(observations) if not isinstance(observations, torch.Tensor) else observations,
                            device=self.model.params.device).to(torch.uint8 if self.frames else torch.float32).reshape(-1, self.obs_dim).contiguous()
        m, A, gaussian = X.shape[0], self.model.action_dim, self.model.dist == "gaussian"
        critic = self.model.head_ld > A
        heads = self.model.forward(X, m, keep=False) if self.frames else self.model.forward(X, m)
        ls = None
        if gaussian:
            ls = self.model.params.ptr(getattr(self.model, "log_std_name", "actor.log_std"))
        if deterministic:
            out = heads[:m, :A]
            acts = out.clone() if gaussian else out.argmax(-1).to(torch.float32)
            logp = None
            if return_logpi:
                logp = torch.log_softmax(out, -1).gather(1, acts.long()[:, None])[:, 0] if not gaussian else None
        else:
            acts = torch.zeros((m, A) if gaussian else (m,), device=X.device)
            logp = torch.zeros(m, device=X.device)
            ops.policy_sample(heads=heads, log_std=ls, act_out=acts, val_out=None, logp_out=logp, env_action=None,
                              env_action_f=None, bootv_prev=None, n=m, A=A, ld=self.model.head_ld, gaussian=int(gaussian),
                              seed=self.seed, step=self._act_calls, step_dev=self.step_counter)
            self._act_calls = (self._act_calls + 1) & 0x7fffffff
        values = heads[:m, A].cpu().numpy() if critic else 0
        env_actions = acts.cpu().numpy() if gaussian else acts.cpu().numpy().astype(np.int64)
        dists = None
        if return_dists:
            dists = {"mu": heads[:m, :A].cpu().numpy(), "log_std": self.model.state_dict()[getattr(self.model, "log_std_name", "actor.log_std")].cpu().numpy()} \
                if gaussian else {"logits": heads[:m, :A].cpu().numpy()}
        return ActionOutput(env_actions=env_actions, values=values, distributions=dists,
                            log_probs=logp.cpu().numpy() if (return_logpi and logp is not None) else None)

    def _process_observation(self, observations, update=False):
        """obs_rms.update (optional) + clip((obs - mean) / (std + 1e-8)) (agent.py:262-283) on the CURRENT statistics, for
        observations that come from outside the device loop (evaluation envs).  Returns a device tensor [m, obs_dim]."""
        X = torch.as_tensor(np.asarray(observations), device=self.model.params.device).to(torch.float32).reshape(-1, self.obs_dim).contiguous()
        if not self.use_obsnorm:
            return X
        mean, var, count = self._obs_stats_tensors()
        out = torch.empty_like(X)
        ops.obs_normalize(x=X, mean=mean, var=var, count=count, out0=out, out1=None, n=X.shape[0], D=self.obs_dim, ld_x=self.obs_dim,
                          ld0=self.obs_dim, ld1=self.obs_dim, update=int(update), normalize=1, range=float(self.obsnorm_range))
        return out

    def _test_actions(self, obs, deterministic):
        # on_policy.py:355-357: the reference updates obs_rms with the evaluation observations too
        return self.get_actions(self._process_observation(obs, update=True), deterministic=deterministic).env_actions


class A2C_Agent(PPO_Agent):
    """Advantage actor-critic agent (xuance/torch/agents/policy_gradient/a2c_agent.py:18-79): the on-policy loop of
    PPO_Agent with A2C_Learner, no auxiliary buffer fields the learner reads, and the reference's ActorCritic model
    (actor and critic each own a copy of the representation, :41-77) -- here one trunk per head, i.e. the
    representation's layers prepended to each head's hidden layers."""

    def _build_model(self):
        c = self.config
        discrete = is_discrete(self.action_space)
        rep = list(_get(c, "representation_hidden_size", []) or []) if _get(c, "representation", "Basic_MLP") != "Basic_Identical" else []
        return ActorCriticNet(self.obs_dim, self.action_space.n if discrete else int(self.action_space.shape[0]),
                              "categorical" if discrete else "gaussian", [], rep + list(c.actor_hidden_size),
                              rep + list(c.critic_hidden_size), _get(c, "activation", "leaky_relu"),
                              None if discrete else _get(c, "activation_action", "tanh"), device=self.device, head_rep_layers=len(rep))

    def _build_learner(self, *args):
        from ..learners.ppo_learner import A2C_Learner
        return A2C_Learner(*args)


class PPOKL_Agent(PPO_Agent):
    """PPO with a KL penalty (xuance/torch/agents/policy_gradient/ppokl_agent.py:7-90): PPO_Agent's loop with PPOKL_Learner;
    what the reference stores per transition as `aux_info = {"old_dist": policy_output.distributions}` (:20-30, a Python
    object per sample) is here the distribution's parameters in two more buffer fields -- old_a [A] (logits / mu) and, for
    Gaussian policies, old_b [A] (std) -- written by the vector step and gathered with the minibatch."""

    def __init__(self, config, envs, callback=None):
        config.use_fused_rollout = False                        # the fused rollout kernels store old_logp only
        super().__init__(config, envs, callback)

    @property
    def auxiliary_info_shape(self):
        discrete = is_discrete(self.action_space)
        A = self.action_space.n if discrete else int(self.action_space.shape[0])
        shape = {"old_logp": (), "old_a": (A,)}
        if not discrete:
            shape["old_b"] = (A,)
        return shape

    def _build_learner(self, *args):
        from ..learners.ppo_learner import PPOKL_Learner
        return PPOKL_Learner(*args)

    def _enqueue_step(self, t):
        super()._enqueue_step(t)
        n, A, f = self.n_envs, self.model.action_dim, self.memory.soa.fields
        heads = self.model.plan.acts[len(self.model.plan.widths) - 1]      # rows [0, n): this step's policy output
        torch.mul(heads[:n, :A], 1.0, out=f["aux_old_a"][t].view(n, A))      # (a kernel, not a memcpy node: graphs)
        if self.model.dist == "gaussian":
            torch.exp(self.model.params.view("actor.log_std", self.model.params.flat).view(1, A).expand(n, A),
                      out=f["aux_old_b"][t].view(n, A))


class PG_Agent(PPO_Agent):
    """Vanilla policy-gradient agent (xuance/torch/agents/policy_gradient/pg_agent.py:12-79 on core/on_policy.py:225-300):
    the on-policy loop with the actor-only VanillaPolicyGradient model (nets.ActorNet) and PG_Learner.  What differs from
    PPO_Agent's loop, as in the reference: the stored value of every step is 0 (on_policy.py:160 `values = 0 if values is
    None`), and the "value" that closes a truncated or buffer-cut path is the PROCESSED REWARD of that step
    (pg_agent.py:66-79 get_terminated_values returns `_process_reward(rewards)`); configs/pg/*.yaml set use_gae: False,
    so returns are discounted reward sums and advantages = returns."""

    def __init__(self, config, envs, callback=None):
        config.use_fused_rollout = False                        # the fused rollout kernels evaluate an actor-critic
        super().__init__(config, envs, callback)

    def _build_model(self):
        from ..nets import ActorNet
        c = self.config
        discrete = is_discrete(self.action_space)
        rep = list(_get(c, "representation_hidden_size", []) or []) if _get(c, "representation", "Basic_MLP") != "Basic_Identical" else []
        return ActorNet(self.obs_dim, self.action_space.n if discrete else int(self.action_space.shape[0]),
                        "categorical" if discrete else "gaussian", rep, list(c.actor_hidden_size),
                        _get(c, "activation", "leaky_relu"), None if discrete else _get(c, "activation_action", "tanh"),
                        device=self.device)

    def _build_learner(self, *args):
        from ..learners.ppo_learner import PG_Learner
        return PG_Learner(*args)

    def _enqueue_step(self, t):
        env, mem, n, D, A = self.envs, self.memory, self.n_envs, self.obs_dim, self.model.action_dim
        f = mem.soa.fields
        gaussian = self.model.dist == "gaussian"
        ops.obs_normalize(x=env.buf_obs, mean=self.obs_mean, var=self.obs_var, count=self.obs_count, out0=self.X,
                          out1=f["observations"][t], n=n, D=D, ld_x=D, ld0=D, ld1=D, update=int(self.use_obsnorm),
                          normalize=int(self.use_obsnorm), range=float(self.obsnorm_range))
        heads = self.model.forward(self.X, n)
        ops.policy_sample(heads=heads, log_std=self.model.params.ptr(self.model.log_std_name) if gaussian else None,
                          noise=None if self.action_noise is None else self.action_noise[t],
                          act_out=f["actions"][t], val_out=None, logp_out=f["aux_old_logp"][t],
                          env_action=None if gaussian else env.action, env_action_f=env.action if gaussian else None,
                          bootv_prev=None, n=n, A=A, ld=self.model.head_ld, gaussian=int(gaussian),
                          seed=self.seed, step=t, step_dev=self.step_counter)
        if hasattr(env, "advance"):
            env.step_device(offset=t)
        else:
            env.step_device()
        # get_terminated_values = the processed reward (pg_agent.py:66-79) over the return statistics of the moment the path closes:
        # written by the same launch (xrl_poststep_t.pg_bootv); terminated envs close with 0 (seg bit 2)
        ops.rollout_poststep(reward=env.reward, terminated=env.terminated, truncated=env.truncated, next_obs=env.next_obs,
                             obs_mean=self.obs_mean, obs_var=self.obs_var, next_obs_norm=self.X[n:], rew_out=f["rewards"][t],
                             term_out=f["terminals"][t], seg_out=f["seg"][t], ret_track=self.returns,
                             ret_mean=self.ret_mean, ret_var=self.ret_var, ret_count=self.ret_count, n=n, D=D, ld_next=D,
                             use_obsnorm=int(self.use_obsnorm), use_rewnorm=int(self.use_rewnorm),
                             last_step=int(t == self.horizon_size - 1), obs_range=float(self.obsnorm_range),
                             rew_range=float(self.rewnorm_range), gamma=float(self.gamma), pg_bootv=f["bootv"][t])

    def _enqueue_rollout(self):
        T = self.horizon_size
        for t in range(T):
            self._enqueue_step(t)
        if hasattr(self.envs, "advance"):
            self.envs.advance(T)
        ops.counter_add(self.step_counter, T)
        f = self.memory.soa.fields
        ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                     self.gamma, self.gae_lam, self.memory.use_gae)

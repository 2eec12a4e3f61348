"""Learner base: config plumbing, optimiser / scheduler facades and checkpoint format of the reference
(xuance/torch/learners/base/drl_learner.py:12-214)."""
import os
from collections import OrderedDict

import torch

from .. import ops


class _NullCallback:
    def on_update_start(self, iterations, **kwargs):
        return {}

    def on_update_end(self, iterations, **kwargs):
        return {}


class AdamHandle:
    """Presents the fused device optimiser (xrl_adam_step) with torch.optim.Adam's state_dict() layout so the
    reference's checkpoint code (drl_learner.py:64-93) and ``optimizer.state_dict()['param_groups'][0]['lr']``
    (ppo_learner.py:69) keep working."""

    def __init__(self, params, ref_order, lr, eps=1e-5, weight_decay=0.0, total_iters=1, end_factor=1.0):
        self.params, self.ref_order = params, list(ref_order)
        self.grad = params.like()
        self.m = params.like()
        self.v = params.like()
        self.state = ops.adam_state_tensor(lr, total_iters, end_factor, eps, weight_decay, device=params.device)
        self.initial_lr = float(lr)       # what torch's scheduler records as param_group['initial_lr']
        # every key torch.optim.Adam keeps in a param group: a file written here loads into the reference's optimiser
        self.defaults = dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                             foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)

    def read(self):
        return ops.read_adam_state(self.state)

    @property
    def lr(self):
        return self.read().last_lr

    def zero_grad(self):
        self.grad.zero_()

    def state_dict(self):
        s = self.read()
        state = {}
        for i, n in enumerate(self.ref_order):
            state[i] = {"step": torch.tensor(float(s.step)), "exp_avg": self.params.view(n, self.m).clone(),
                        "exp_avg_sq": self.params.view(n, self.v).clone()}
        group = dict(self.defaults, lr=s.last_lr, initial_lr=self.initial_lr, params=list(range(len(self.ref_order))))
        return {"state": state if s.step > 0 else {}, "param_groups": [group]}

    def load_state_dict(self, sd):
        """torch.optim.Adam.load_state_dict as the reference's resume sees it (drl_learner.py:124-135): the moments and
        step counts come back, and so does the DECAYED param_group lr -- while the scheduler object of the new process
        is fresh.  torch's chained LinearLR then continues as lr_saved * (1 + (end_factor - 1) * j / total_iters), j =
        scheduler steps since the resume (the chained factors telescope; fixture tests/golden/ppo_ckpt_decay.npz).  The
        device computes lr = base_lr * (1 + (end_factor - 1) * min(sched_steps, total) / total), so exactly that is
        base_lr := the restored lr, sched_steps := 0; `initial_lr` is only reported back in state_dict()."""
        s = self.read()
        for i, n in enumerate(self.ref_order):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.params.view(n, self.m).copy_(st["exp_avg"])
            self.params.view(n, self.v).copy_(st["exp_avg_sq"])
            s.step = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.initial_lr = float(g.get("initial_lr", g["lr"]))
        s.base_lr = float(g["lr"])
        s.last_lr = float(g["lr"])
        s.sched_steps = 0
        ops.write_adam_state(self.state, s)
        self.generation = getattr(self, "generation", 0) + 1        # users with launch-epoch barriers re-arm them


class LinearLRHandle:
    """torch.optim.lr_scheduler.LinearLR(start_factor=1.0, end_factor, total_iters) facade (ppo_learner.py:19-22)."""

    def __init__(self, opt: AdamHandle):
        self.opt = opt

    @property
    def last_epoch(self):
        return self.opt.read().sched_steps

    def get_last_lr(self):
        return [self.opt.read().last_lr]

    def state_dict(self):
        s = self.opt.read()
        return dict(start_factor=1.0, end_factor=s.end_factor, total_iters=s.total_iters, last_epoch=s.sched_steps,
                    _last_lr=[s.last_lr])

    def step(self, epoch=None):
        """Stepping happens inside the Adam launches; this is the host-side hook of _safe_scheduler_step
        (drl_learner.py:189-210).  step(epoch): torch's closed form, lr = scheduler.base_lrs * factor(epoch), where
        base_lrs is the learning rate the scheduler was CONSTRUCTED with (the config value), not the restored one."""
        s = self.opt.read()
        if epoch is None:
            s.sched_steps += 1
        else:
            s.sched_steps = int(epoch)
            s.base_lr = float(self.opt.defaults["lr"])
        k = min(s.sched_steps, s.total_iters)
        s.last_lr = s.base_lr * (1.0 + (s.end_factor - 1.0) * k / s.total_iters)
        ops.write_adam_state(self.opt.state, s)


class Learner:
    def __init__(self, config, model, callback=None, adopt_hints=None):
        """`model`: a xuance_amd.nets container, or the policy object the REFERENCE built (an nn.Module such as
        SharedActorCritic / DeepQNetwork / MixingQNetwork -- what Agent._build_learner hands to
        REGISTRY_Learners[config.learner], agent.py:340-341): it is adopted (xuance_amd/adapters.py), i.e. rebuilt over the
        flat device buffers with the module's parameters re-pointed at them, so module and engine share storage."""
        from ..adapters import adopt
        self.config = config
        model = adopt(model, config, **(adopt_hints or {}))
        self.distributed_training = getattr(config, "distributed_training", False)
        self.episode_length = getattr(config, "episode_length", None)
        self.learning_rate = getattr(config, "learning_rate", None)
        self.end_factor_lr_decay = getattr(config, "end_factor_lr_decay", 1.0)
        self.gamma = getattr(config, "gamma", 0.99)
        self.use_rnn = getattr(config, "use_rnn", False)
        self.use_actions_mask = getattr(config, "use_actions_mask", False)
        self.model = model
        self.policy = getattr(model, "module", model)           # what callbacks receive: the caller's own object
        self.optimizer = None
        self.scheduler = None
        self.callback = callback if callback is not None else _NullCallback()
        if self.distributed_training:
            self.world_size = int(os.environ["WORLD_SIZE"])
            self.rank = int(os.environ["RANK"])
            if self.world_size > 1:                             # what Agent.__init__ does in the reference (agent.py:76-80:
                import torch.distributed as tdist               # init_distributed_mode) when nobody has done it yet
                if not tdist.is_initialized():
                    from ..dist import init_distributed_mode
                    init_distributed_mode()
        else:
            self.world_size, self.rank = 1, 0
        self.use_grad_clip = getattr(config, "use_grad_clip", False)
        self.grad_clip_norm = getattr(config, "grad_clip_norm", 0.5)
        self.device = getattr(config, "device", "cuda")
        self.model_dir = getattr(config, "model_dir", "models")
        self.snapshot_path = os.path.join(os.getcwd(), self.model_dir, "DDP_Snapshot")
        self.total_iters = self.estimate_total_iterations()
        self.iterations = 0

    def gradient_exchange(self):
        """The ranks' gradient exchange buffers (dist.GradientExchange) when averaging inside the optimiser launch is
        usable in this job, else None (the learners then all-reduce through the process group).  Collective on first use:
        every rank gets here at the same point of its first update."""
        if not hasattr(self, "_xc"):
            from .. import dist as xdist
            self._xc = None
            from .. import ops
            if self.distributed_training and self.world_size > 1 and getattr(self.config, "dist_gradient_exchange", True) \
                    and self.model.params.P % 4 == 0 and (self.model.params.P + 255) // 256 <= ops.XC_MAX_GROUPS \
                    and xdist.exchange_usable(self.device):
                self._xc = xdist.GradientExchange(self.model.params.P, self.device)
        return self._xc

    OPT_TIMEOUT_MSG = ("xrl_reduce_adam: %s timed out -- the optimiser step of this update phase is invalid: parameters of the "
                       "blocks that saw the time-out were NOT stepped, later launches of the phase stepped nothing -- the replica is "
                       "partially updated and cannot be repaired in place: reload a checkpoint (set use_fused_optimizer: False to use "
                       "the two-launch sequence, dist_gradient_exchange: False to average through the process group)")

    def raise_on_optimizer_timeout(self, code=None):
        """xrl_reduce_adam's status word sync[2]: 1 = the inter-block barrier, 2 = the wait for the other ranks' gradient rows
        expired.  Every learner calls this where it reads an update's results back (`code` given: the caller has the word
        already, e.g. inside its own read-back; else one 4-byte copy per optimiser scratch tensor of this learner)."""
        if code is None:
            code = 0
            for name in ("opt_sync", "_lsync"):
                t = getattr(self, name, None)
                if t is not None:
                    code = code or int(t[2].item())
        if code:
            raise ops.XrlError(self.OPT_TIMEOUT_MSG % ("the wait for the other ranks' gradient rows" if code == 2
                                                       else "the inter-block barrier"))

    def read_optimizer(self):
        """optimizer.read() + the time-out check (the host is synchronous here anyway)."""
        st = self.optimizer.read()
        self.raise_on_optimizer_timeout()
        return st

    def _fused_optimizer_ok(self, with_exchange=False):
        """May this learner's optimiser step be ONE launch (xrl_reduce_adam: slab reduction, norm, clip, Adam, derived layouts behind
        a spinning inter-block barrier)?  Not switched off (config.use_fused_optimizer) and every block of that launch resident on
        this device (xrl_reduce_adam_fits); else the two-launch sequence xrl_grad_reduce + xrl_adam_step runs -- same numbers."""
        key = "_foo_x" if with_exchange else "_foo"
        if not hasattr(self, key):
            setattr(self, key, bool(getattr(self.config, "use_fused_optimizer", True)) and
                    ops.reduce_adam_fits(self.model.params.P, with_exchange))
        return getattr(self, key)

    def needs_collective(self):
        """Several ranks AND no in-launch exchange: the update must stop at a process-group all-reduce."""
        return bool(self.distributed_training and self.world_size > 1 and self.gradient_exchange() is None)

    def sync_replicas_from_rank0(self):
        """What the reference's DistributedDataParallel wrap does at construction (deep_q_network.py:55-59,
        value_factorization.py:44-48): every rank starts from rank 0's parameters (and target copies).  Ranks are seeded
        differently on purpose -- distinct env shards, distinct action draws -- so without this they would average
        gradients across diverged replicas.  Called at the end of every learner constructor; the derived parameter
        layouts of the fused kernels are (re)built from params.flat at the start of every update phase anyway."""
        if not (self.distributed_training and self.world_size > 1):
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            if os.environ.get("XRL_DIST_STUB") == "1":          # tools/time_distributed_path.py: structure cost, no peers
                return
            raise RuntimeError("distributed_training with WORLD_SIZE > 1 but torch.distributed is not initialised: the ranks "
                               "would train independent replicas (xuance_amd.dist.init_distributed_mode() -- the agents call it)")
        from ..dist import broadcast_
        broadcast_(self.model.params.flat, 0)
        if getattr(self.model, "target_flat", None) is not None:
            broadcast_(self.model.target_flat, 0)

    def estimate_total_iterations(self):                        # drl_learner.py:56-62
        start_training = getattr(self.config, "start_training", 0)
        training_frequency = getattr(self.config, "training_frequency", 1)
        return (self.config.running_steps - start_training) // (training_frequency * self.config.parallels)

    def _key(self, k):                                          # "/rank_{r}" suffix (ppo_learner.py:72-80)
        return f"{k}/rank_{self.rank}" if self.distributed_training else k

    # -- checkpoint format of drl_learner.py:64-157 ------------------------------------------------------
    def save_model(self, model_path):
        os.makedirs(os.path.dirname(model_path) or ".", exist_ok=True)
        osd = self.optimizer.state_dict()
        osd["state"] = {i: {k: v.cpu() for k, v in st.items()} for i, st in osd["state"].items()}
        torch.save({"policy": OrderedDict((k, v.cpu()) for k, v in self.model.state_dict().items()),
                    "optimizer": osd,
                    "rng_state": torch.get_rng_state(),
                    "cuda_rng_state": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else []},
                   model_path)

    def load_model(self, path, model=None):
        """drl_learner.py:95-157: `path`/`model` names a file, or `path` is a directory holding `seed_*` run folders (the
        last one in sorted order is taken, `final_train_model.pth` preferred inside it) -- or, as a convenience, `.pth`
        files directly.  Returns the DIRECTORY the file was loaded from (the agent looks for obs_rms.npy there)."""
        target = os.path.join(path, model) if model is not None else path
        if os.path.isfile(target):
            path = target
        else:
            if not os.path.isdir(path):
                raise RuntimeError(f"The path '{path}' is not a valid directory or file!")
            runs = sorted(f for f in os.listdir(path) if "seed_" in f and os.path.isdir(os.path.join(path, f)))
            folder = os.path.join(path, runs[-1]) if runs else path
            files = sorted(f for f in os.listdir(folder) if f.endswith(".pth"))
            if not files:
                raise (FileNotFoundError(f"No .pth file found in {folder}") if runs else
                       RuntimeError(f"No model files with 'seed_' found in '{path}'!"))
            path = os.path.join(folder, "final_train_model.pth" if "final_train_model.pth" in files else files[-1])
        ckpt = torch.load(path, map_location="cpu", weights_only=True)          # drl_learner.py:119-121
        self.model.load_state_dict(ckpt["policy"] if "policy" in ckpt else ckpt)
        if "optimizer" in ckpt and self.optimizer is not None:
            self.optimizer.load_state_dict(ckpt["optimizer"])
            self.learning_rate = self.optimizer.lr                              # :133-135
        if ckpt.get("rng_state") is not None:                                   # :137-140
            torch.set_rng_state(ckpt["rng_state"].cpu().to(torch.uint8))
        if ckpt.get("cuda_rng_state") and torch.cuda.is_available():           # :142-153
            for i, st in enumerate(ckpt["cuda_rng_state"][:torch.cuda.device_count()]):
                torch.cuda.set_rng_state(st.cpu().to(torch.uint8), device=i)
        self._safe_scheduler_step()
        if hasattr(self.model, "_act_state"):                                   # derived weight images follow on their next use
            self.model._act_stale = True
        self._images_current = False
        self._images_stale = True
        if getattr(self, "_xc", None) is not None:                              # so do the exchange buffers' flags (all ranks load)
            self._xc.clear()
        for name in ("opt_sync", "_lsync"):                                     # barrier flags of xrl_reduce_adam hold step
            if getattr(self, name, None) is not None:                           # values: a rewound step must not match them
                getattr(self, name).zero_()
        return os.path.dirname(path)

    def _safe_scheduler_step(self):
        """drl_learner.py:189-210: only when the config carries `rt_epoch` (resumed benchmark runs) the scheduler jumps to
        the iteration that epoch corresponds to."""
        if self.scheduler is None or not hasattr(self.config, "rt_epoch"):
            return
        train_steps = self.config.running_steps // self.config.parallels
        eval_interval = self.config.eval_interval // self.config.parallels
        num_epoch = int(train_steps / eval_interval)
        self.scheduler.step(int(self.total_iters * self.config.rt_epoch / num_epoch))

    def update(self, *args, **kwargs):
        raise NotImplementedError

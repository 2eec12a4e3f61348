"""HBM-resident rollout / replay buffers with the reference Buffer surface
(``store / finish_path / sample / clear / full / ptr / size / n_size / n_envs / buffer_size``,
xuance/common/memory_tools.py:87-142) over a TIME-MAJOR structure of arrays ``field[t][env][...]``.

  HipOnPolicyBuffer        <-> DummyOnPolicyBuffer(_Atari)   memory_tools.py:182-328
  HipOffPolicyBuffer       <-> DummyOffPolicyBuffer(_Atari)  memory_tools.py:331-387, 601-630

All fields of a buffer are carved out of one device allocation; the per-step write is one contiguous copy per
field (xrl_soa_store_step), GAE for every env and every path is one launch (xrl_gae_scan) and sampling is one
gather launch (xrl_soa_gather).  Returned samples are device tensors: the learners (ours and the reference's,
which wrap inputs in torch.as_tensor(..., device=...)) consume them without a host round trip.
"""
import numpy as np
import torch

from . import ops
from .spaces import space2shape


def _nbytes(shape, itemsize):
    n = itemsize
    for s in shape:
        n *= s
    return n


class _SoA:
    """Fields [T][n_envs][row] carved from one uint8 device allocation (16-byte aligned each)."""

    def __init__(self, T, n_envs, specs, device):
        self.T, self.n_envs, self.device = T, n_envs, device
        self.specs = specs                                   # name -> (shape, torch dtype)
        off, self.offsets = 0, {}
        for name, (shape, dtype) in specs.items():
            nb = T * n_envs * _nbytes(shape, torch.empty((), dtype=dtype).element_size())
            self.offsets[name] = (off, nb)
            off = (off + nb + 255) // 256 * 256
        self.storage = torch.zeros(off, dtype=torch.uint8, device=device)
        self.fields = {}
        for name, (shape, dtype) in specs.items():
            o, nb = self.offsets[name]
            self.fields[name] = self.storage[o:o + nb].view(dtype).view((T, n_envs) + tuple(shape))
        self.row_bytes = {n: _nbytes(s, torch.empty((), dtype=d).element_size()) for n, (s, d) in specs.items()}

    def zero(self):
        self.storage.zero_()


class _Stager:
    """Packs the per-step host arrays into ONE pinned buffer -> one H2D copy -> device step tensors."""

    def __init__(self, n_envs, specs, device):
        self.views_h, self.views_d = {}, {}
        self._evt = None
        off, lay = 0, {}
        for name, (shape, dtype) in specs.items():
            nb = n_envs * _nbytes(shape, torch.empty((), dtype=dtype).element_size())
            lay[name] = (off, nb)
            off = (off + nb + 15) // 16 * 16
        self.host = torch.zeros(max(off, 16), dtype=torch.uint8)
        if torch.cuda.is_available():
            self.host = self.host.pin_memory()
        self.dev = torch.zeros(max(off, 16), dtype=torch.uint8, device=device)
        for name, (shape, dtype) in specs.items():
            o, nb = lay[name]
            self.views_h[name] = self.host[o:o + nb].view(dtype).view((n_envs,) + tuple(shape))
            self.views_d[name] = self.dev[o:o + nb].view(dtype).view((n_envs,) + tuple(shape))

    def put(self, items):
        """items: name -> numpy array / scalar / tensor.  Returns name -> device step tensor."""
        out, need_copy = {}, False
        if self._evt is not None:
            self._evt.synchronize()           # the previous async H2D copy must have consumed the pinned buffer
        for name, x in items.items():
            if isinstance(x, torch.Tensor) and x.is_cuda:
                v = self.views_d[name]
                out[name] = x.to(v.dtype).reshape(v.shape).contiguous()
            else:
                h = self.views_h[name]
                h.copy_(torch.as_tensor(np.asarray(x)).to(h.dtype).reshape(h.shape))
                out[name] = self.views_d[name]
                need_copy = True
        if need_copy:
            self.dev.copy_(self.host, non_blocking=True)
            if self.dev.is_cuda:
                self._evt = torch.cuda.Event()
                self._evt.record()
        return out


class HipOnPolicyBuffer:
    """Drop-in for DummyOnPolicyBuffer / TensorOnPolicyBuffer (same constructor keywords, tensor_memory.py:165-175)."""

    def __init__(self, observation_space, action_space, auxiliary_shape, n_envs, horizon_size, use_gae=True,
                 use_advnorm=True, gamma=0.99, gae_lam=0.95, device="cuda", obs_dtype=torch.float32):
        self.observation_space, self.action_space, self.auxiliary_shape = observation_space, action_space, auxiliary_shape
        self.n_envs, self.horizon_size, self.n_size = n_envs, horizon_size, horizon_size
        self.buffer_size = horizon_size * n_envs
        self.use_gae, self.use_advnorm, self.gamma, self.gae_lam = use_gae, use_advnorm, gamma, gae_lam
        self.device = device
        self.obs_shape, self.act_shape = space2shape(observation_space), space2shape(action_space)
        f32 = torch.float32
        specs = {"observations": (self.obs_shape, obs_dtype), "actions": (self.act_shape, f32), "rewards": ((), f32),
                 "returns": ((), f32), "values": ((), f32), "terminals": ((), f32), "advantages": ((), f32),
                 "bootv": ((), f32), "seg": ((), torch.uint8)}
        self.aux_keys = list(auxiliary_shape.keys()) if auxiliary_shape else []
        for k in self.aux_keys:
            specs["aux_" + k] = (tuple(auxiliary_shape[k]), f32)
        self.soa = _SoA(self.n_size, n_envs, specs, device)
        step_specs = {k: specs[k] for k in ["observations", "actions", "rewards", "values", "terminals"]}
        for k in self.aux_keys:
            step_specs["aux_" + k] = specs["aux_" + k]
        self.stager = _Stager(n_envs, step_specs, device)
        self.start_ids = np.zeros(n_envs, np.int64)
        self._seg_h = np.zeros((self.n_size, n_envs), np.uint8)
        self._bootv_h = np.zeros((self.n_size, n_envs), np.float32)
        self._stats = torch.zeros(2, device=device)
        self.ptr, self.size = 0, 0
        self._dirty = False

    # -- reference surface -----------------------------------------------------------------------------
    @property
    def full(self):
        return self.size >= self.n_size

    def clear(self):                                          # memory_tools.py:221-230
        self.ptr, self.size = 0, 0
        self.soa.zero()
        self._seg_h[:] = 0
        self._bootv_h[:] = 0
        self._dirty = False

    def store(self, obs, acts, rews, value, terminals, aux_info=None):      # memory_tools.py:232-240
        items = {"observations": obs, "actions": acts, "rewards": rews, "values": value, "terminals": terminals}
        for k in self.aux_keys:
            if aux_info is not None and k in aux_info:
                items["aux_" + k] = aux_info[k]
        step = self.stager.put(items)
        f = self.soa
        ops.soa_store_step([(f.fields[k], step[k], f.row_bytes[k]) for k in step], self.n_envs, self.ptr)
        self.ptr = (self.ptr + 1) % self.n_size
        self.size = min(self.size + 1, self.n_size)
        self._dirty = True

    def finish_path(self, val, i):                            # memory_tools.py:242-265
        """Records that env i's current path ends here with bootstrap value ``val``; the scan itself is deferred
        to one xrl_gae_scan launch over all envs (run before anything reads returns / advantages)."""
        end = self.n_size if self.full else self.ptr
        start = int(self.start_ids[i])
        if end > start:
            if isinstance(val, torch.Tensor):
                val = val.detach().cpu().numpy()
            py_float = isinstance(val, float) and not isinstance(val, np.floating)   # Python float -> float64 carry
            if start == 0 and self._seg_h[:end, i].any():
                # a second call over the whole row (full buffer, start_ids wrapped to 0 without a clear()):
                # the reference recomputes [0, n_size) as ONE path, dropping the earlier path boundaries.
                self._seg_h[:end, i] = 0
            self._seg_h[end - 1, i] = 1 | (2 if py_float else 0)
            self._bootv_h[end - 1, i] = np.float32(val)
            self._dirty = True
        self.start_ids[i] = self.ptr

    def finish_paths(self, vals, terminated=None):
        """Vector form of the buffer-full loop of ppo_agent.py:129-135: one call for every env."""
        vals = np.asarray(vals.detach().cpu().numpy() if isinstance(vals, torch.Tensor) else vals, np.float32)
        for i in range(self.n_envs):
            if terminated is not None and terminated[i]:
                self.finish_path(0.0, i)
            else:
                self.finish_path(vals[i], i)

    def _sync(self):
        if not self._dirty:
            return
        f = self.soa.fields
        f["seg"].copy_(torch.from_numpy(self._seg_h))
        f["bootv"].copy_(torch.from_numpy(self._bootv_h))
        ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                     self.gamma, self.gae_lam, self.use_gae)
        self._dirty = False

    def sample(self, indexes):                                # memory_tools.py:267-287
        assert self.full, "Not enough transitions for on-policy buffer to random sample"
        self._sync()
        idx = torch.as_tensor(np.asarray(indexes) if not isinstance(indexes, torch.Tensor) else indexes)
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        bs = idx.numel()
        f, dev = self.soa, self.device
        out = {"observations": torch.empty((bs,) + self.obs_shape, dtype=f.fields["observations"].dtype, device=dev),
               "actions": torch.empty((bs,) + self.act_shape, device=dev), "returns": torch.empty(bs, device=dev),
               "values": torch.empty(bs, device=dev), "advantages": torch.empty(bs, device=dev)}
        for k in self.aux_keys:
            out["aux_" + k] = torch.empty((bs,) + tuple(self.auxiliary_shape[k]), device=dev)
        names = list(out)
        if self.use_advnorm:
            ops.adv_stats(f.fields["advantages"], idx, bs, 1, self.n_envs, self.n_size, self._stats)
        flags = [1 if (n == "advantages" and self.use_advnorm) else 0 for n in names]
        ops.soa_gather([(out[n], f.fields[n], f.row_bytes[n]) for n in names], idx, self.n_envs, self.n_size,
                       stats=self._stats if self.use_advnorm else None, flags=flags)
        return {"obs": out["observations"], "actions": out["actions"], "returns": out["returns"],
                "values": out["values"], "aux_batch": {k: out["aux_" + k] for k in self.aux_keys},
                "batch_size": bs, "advantages": out["advantages"]}

    # -- views in the reference's env-major orientation (for inspection / tests) ----------------------------
    def field(self, name):
        self._sync()
        return self.soa.fields[name]

    @property
    def returns(self):
        return self.field("returns").transpose(0, 1)

    @property
    def advantages(self):
        return self.field("advantages").transpose(0, 1)


class HipOnPolicyBuffer_Atari(HipOnPolicyBuffer):
    """uint8 observations (memory_tools.py:290-328)."""

    def __init__(self, *args, **kwargs):
        kwargs["obs_dtype"] = torch.uint8
        super().__init__(*args, **kwargs)


class HipOffPolicyBuffer:
    """Drop-in for DummyOffPolicyBuffer (memory_tools.py:331-387): ring [n_size][n_envs] per field in HBM."""

    def __init__(self, observation_space, action_space, auxiliary_shape, n_envs, buffer_size, batch_size,
                 device="cuda", obs_dtype=torch.float32):
        assert buffer_size % n_envs == 0, "buffer_size must be divisible by the number of envs (parallels)"
        self.observation_space, self.action_space = observation_space, action_space
        self.n_envs, self.buffer_size, self.batch_size = n_envs, buffer_size, batch_size
        self.n_size = buffer_size // n_envs
        self.device = device
        self.obs_shape, self.act_shape = space2shape(observation_space), space2shape(action_space)
        f32 = torch.float32
        specs = {"observations": (self.obs_shape, obs_dtype), "next_observations": (self.obs_shape, obs_dtype),
                 "actions": (self.act_shape, f32), "rewards": ((), f32), "terminals": ((), f32)}
        self.soa = _SoA(self.n_size, n_envs, specs, device)
        self.stager = _Stager(n_envs, specs, device)
        self.ptr, self.size = 0, 0
        self.size_dev = torch.zeros(1, dtype=torch.int32, device=device)

    @property
    def full(self):
        return self.size >= self.n_size

    def clear(self):
        self.ptr, self.size = 0, 0
        self.size_dev.zero_()
        self.soa.zero()

    def store(self, obs, acts, rews, terminals, next_obs):     # memory_tools.py:365-372
        step = self.stager.put({"observations": obs, "actions": acts, "rewards": rews, "terminals": terminals,
                                "next_observations": next_obs})
        f = self.soa
        grow = self.size < self.n_size                         # (the filled-slot count for device-side sampling rides in the launch)
        ops.soa_store_step([(f.fields[k], step[k], f.row_bytes[k]) for k in step], self.n_envs, self.ptr,
                           size_dev=self.size_dev if grow else None, new_size=self.size + 1)
        self.ptr = (self.ptr + 1) % self.n_size
        if grow:
            self.size += 1                   # `size` for sampling kernels inside captured graphs

    def fill_synthetic(self, seed=0, chunk=256):
        """Fill the WHOLE ring with synthetic transitions on the device (frames / observations uniform over their dtype's range in
        slabs of `chunk` slots, actions over the action set, N(0,1) rewards, 2 % terminals) and mark it full: measurements at a
        configuration's real replay size (configs/dqn/atari.yaml: 500 000 frames = 28 GB of uint8 stacks) without the
        hours of environment steps that fill it."""
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        f = self.soa.fields
        for k in ("observations", "next_observations"):
            x = f[k]
            for t0 in range(0, self.n_size, chunk):
                if x.dtype == torch.uint8:
                    x[t0:t0 + chunk].random_(0, 256, generator=g)
                else:
                    x[t0:t0 + chunk].normal_(generator=g)
        n_act = getattr(self.action_space, "n", None)
        if n_act is not None:
            f["actions"].random_(0, int(n_act), generator=g)
        else:
            f["actions"].uniform_(-1.0, 1.0, generator=g)
        f["rewards"].normal_(generator=g)
        f["terminals"].copy_((torch.rand(f["terminals"].shape, device=self.device, generator=g) < 0.02).float())
        self.size = self.n_size
        self.size_dev.fill_(self.n_size)

    def ring_bias(self, counter_value):
        """(slot_bias, size_bias) of store_ring for a device counter that currently holds `counter_value`: the next store lands in
        slot `ptr` and makes `size + 1` slots filled."""
        # (a full ring: any bias that keeps min(size_bias + c + 1, n_size) at n_size -- one constant, so that a caller's captured
        #  arguments stop changing once the ring has wrapped)
        return (self.ptr - int(counter_value)) % self.n_size, (self.size - int(counter_value)) if self.size < self.n_size else 1 << 40

    def store_ring(self, obs, acts, rews, terminals, next_obs, counter_dev, offset, bias, mirror=True):
        """store() of device tensors inside a captured vector step: slot and filled-slot count come from `counter_dev` + offset
        (xrl_soa_store_step_ring; `bias` = ring_bias(counter's value at capture)).  The host mirrors advance as in store()."""
        step = {"observations": obs, "actions": acts, "rewards": rews, "terminals": terminals, "next_observations": next_obs}
        f = self.soa
        for k, x in step.items():
            assert x.is_cuda and x.is_contiguous() and x.dtype == f.fields[k].dtype, k
        ops.soa_store_step_ring([(f.fields[k], step[k], f.row_bytes[k]) for k in step], self.n_envs, self.n_size, bias[0], bias[1],
                                counter_dev, offset, self.size_dev)
        if mirror:
            self.advance_mirrors(1)

    def advance_mirrors(self, k):
        """Host-side ptr / size after k stores that ran on the device (a replayed graph)."""
        self.ptr = (self.ptr + k) % self.n_size
        self.size = min(self.size + k, self.n_size)

    def gather_into(self, idx, dst):
        """dst: field name -> device tensor [bs, row] (a learner's staging views); one launch, no host work."""
        f = self.soa
        ops.soa_gather([(dst[k], f.fields[k], f.row_bytes[k]) for k in dst], idx, self.n_envs, self.n_size)

    def draw_into(self, idx_out, dst, seed, counter, counter_dev):
        """Uniform draw (xrl_sample_replay_indices' stream, following the filling ring through size_dev) + gather_into as
        ONE launch; idx_out receives the rows that were picked.  Batches above 256 rows take the two launches."""
        f = self.soa
        if idx_out.numel() > 256:
            ops.sample_replay_indices(idx_out, self.n_envs, self.n_size, self.size_dev, seed, counter, counter_dev)
            return self.gather_into(idx_out, dst)
        ops.soa_gather_sampled([(dst[k], f.fields[k], f.row_bytes[k]) for k in dst], idx_out, self.n_envs, self.n_size,
                               self.size_dev, seed, counter, counter_dev)

    def sample_indices(self, batch_size=None):
        """The two NumPy global-RNG draws of memory_tools.py:376-377, as flat env-major indices."""
        bs = self.batch_size if batch_size is None else batch_size
        env_choices = np.random.choice(self.n_envs, bs)
        step_choices = np.random.choice(self.size, bs)
        return env_choices * self.n_size + step_choices

    def sample(self, batch_size=None, indexes=None):           # memory_tools.py:374-387
        idx = self.sample_indices(batch_size) if indexes is None else indexes
        idx = torch.as_tensor(np.asarray(idx) if not isinstance(idx, torch.Tensor) else idx)
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        bs = idx.numel()
        f, dev = self.soa, self.device
        odt = f.fields["observations"].dtype
        out = {"observations": torch.empty((bs,) + self.obs_shape, dtype=odt, device=dev),
               "actions": torch.empty((bs,) + self.act_shape, device=dev),
               "next_observations": torch.empty((bs,) + self.obs_shape, dtype=odt, device=dev),
               "rewards": torch.empty(bs, device=dev), "terminals": torch.empty(bs, device=dev)}
        ops.soa_gather([(out[n], f.fields[n], f.row_bytes[n]) for n in out], idx, self.n_envs, self.n_size)
        return {"obs": out["observations"], "actions": out["actions"], "obs_next": out["next_observations"],
                "rewards": out["rewards"], "terminals": out["terminals"], "batch_size": bs}


class HipOffPolicyBuffer_Atari(HipOffPolicyBuffer):
    """uint8 frames (memory_tools.py:601-630): 2 x 28 224 B per transition for 84x84x4."""

    def __init__(self, *args, **kwargs):
        kwargs["obs_dtype"] = torch.uint8
        super().__init__(*args, **kwargs)


class HipPerOffPolicyBuffer(HipOffPolicyBuffer):
    """Prioritized replay, drop-in for PerOffPolicyBuffer (memory_tools.py:471-598): the transition ring of
    HipOffPolicyBuffer plus one sum / min segment tree per env in HBM (csrc/per.hip).  `sample(beta)` draws
    batch_size / n_envs stratified transitions per env; its uniforms are Python's `random.random()` in the reference's
    order (one small H2D copy), so a seeded run picks the same transitions as the reference; `update_priorities` keeps the
    reference's in-order semantics per env.  Priorities are float64 (what the reference computes under the NumPy < 2 it
    pins; under NumPy >= 2 its float32 |td| would stay float32 through `**`)."""

    def __init__(self, observation_space, action_space, auxiliary_shape, n_envs, buffer_size, batch_size, alpha=0.6,
                 device="cuda", obs_dtype=torch.float32):
        super().__init__(observation_space, action_space, auxiliary_shape, n_envs, buffer_size, batch_size, device, obs_dtype)
        assert batch_size % n_envs == 0, "PerOffPolicyBuffer draws batch_size / n_envs transitions per env"
        self._alpha = float(alpha)
        self.capacity = 1
        while self.capacity < self.n_size:                      # :499-501
            self.capacity *= 2
        self.per_env = batch_size // n_envs
        self._reset_trees()
        k = self.per_env
        self._uni = torch.zeros(n_envs, k, dtype=torch.float64, device=device)
        self._uni_h = torch.zeros(n_envs, k, dtype=torch.float64)
        if torch.cuda.is_available():
            self._uni_h = self._uni_h.pin_memory()
        self.step_choices = torch.zeros(n_envs, k, dtype=torch.int64, device=device)
        self.weights = torch.zeros(n_envs, k, dtype=torch.float64, device=device)
        self.flat_idx = torch.zeros(n_envs * k, dtype=torch.int64, device=device)

    def _reset_trees(self):
        dev = self.device
        self.it_sum = torch.zeros(self.n_envs, 2 * self.capacity, dtype=torch.float64, device=dev)
        self.it_min = torch.full((self.n_envs, 2 * self.capacity), float("inf"), dtype=torch.float64, device=dev)
        self.max_priority = torch.ones(self.n_envs, dtype=torch.float64, device=dev)

    def clear(self):
        super().clear()
        self._reset_trees()

    def store(self, obs, acts, rews, terminals, next_obs):     # :528-541
        ptr = self.ptr
        super().store(obs, acts, rews, terminals, next_obs)
        ops.per_store(self.it_sum, self.it_min, self.max_priority, ptr, self._alpha, self.n_envs, self.capacity)

    def sample(self, beta, uniforms=None):                     # :542-584
        import random
        assert beta > 0
        if uniforms is None:
            if getattr(self, "_uni_copied", None) is not None:  # the previous asynchronous copy still reads the pinned block
                self._uni_copied.synchronize()
            for i in range(self.n_envs):                        # _sample_proportional's draws, env by env (:504-506)
                for j in range(self.per_env):
                    self._uni_h[i, j] = random.random()
            self._uni.copy_(self._uni_h, non_blocking=True)
            if self._uni_h.is_pinned():
                self._uni_copied = torch.cuda.Event()
                self._uni_copied.record()
        else:
            self._uni.copy_(torch.as_tensor(np.asarray(uniforms), dtype=torch.float64).reshape(self._uni.shape))
        ops.per_sample(self.it_sum, self.it_min, self._uni, self.size, beta, self.n_envs, self.n_size, self.capacity,
                       self.per_env, self.step_choices, self.weights, self.flat_idx)
        out = super().sample(indexes=self.flat_idx)
        out.update(weights=self.weights, step_choices=self.step_choices, batch_size=self.batch_size)
        return out

    def update_priorities(self, idxes, priorities):            # :586-597
        idx = torch.as_tensor(np.asarray(idxes) if not isinstance(idxes, torch.Tensor) else idxes)
        idx = idx.to(device=self.device, dtype=torch.int64).reshape(self.n_envs, self.per_env).contiguous()
        pr = torch.as_tensor(np.asarray(priorities) if not isinstance(priorities, torch.Tensor) else priorities)
        pr = pr.to(device=self.device, dtype=torch.float32).reshape(self.n_envs, self.per_env).contiguous()
        ops.per_update_priorities(self.it_sum, self.it_min, self.max_priority, idx, pr, self._alpha, self.n_envs,
                                  self.capacity, self.per_env)

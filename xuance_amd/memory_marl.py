"""HBM-resident multi-agent replay buffer with the MARL_OffPolicyBuffer surface
(xuance/common/memory_tools_marl.py:634-767): ``store(**step_data)`` takes the reference's nested dicts
(field -> agent -> array [n_envs, ...]) or already stacked tensors, ``sample(batch_size)`` draws uniform
(env, step) pairs with the same two NumPy global-RNG calls and returns the nested per-agent format (views of the
stacked device tensors), so both QMIX_Learner implementations consume it.

Layout: the reference keeps one NumPy array per agent and field, [n_envs][n_size][dim]; under parameter sharing all
agents are homogeneous, so here each field is ONE time-major device array [n_size][n_envs][n_agents*dim]: the per-step
write is one contiguous copy per field and a sampled transition is one contiguous row per field.
The recurrent/episode variant (MARL_OffPolicyBuffer_RNN, :770-996) is SURVEY section 8f "next".
"""
import numpy as np
import torch

from . import ops
from .memory import _SoA, _Stager
from .spaces import space2shape


class HipMARLOffPolicyBuffer:
    def __init__(self, agent_keys, state_space=None, obs_space=None, act_space=None, n_envs=1, buffer_size=1,
                 batch_size=1, device="cuda", **kwargs):
        self.agent_keys = list(agent_keys)
        self.n_agents = len(self.agent_keys)
        assert buffer_size % n_envs == 0, "buffer_size must be divisible by the number of envs (parallels)"
        self.n_envs, self.buffer_size, self.batch_size = n_envs, buffer_size, batch_size
        self.n_size = buffer_size // n_envs
        self.device = device
        self.store_global_state = state_space is not None
        self.use_actions_mask = kwargs.get("use_actions_mask", False)
        k0 = self.agent_keys[0]
        self.obs_dim = int(np.prod(space2shape(obs_space[k0])))
        self.act_shape = space2shape(act_space[k0])
        self.state_dim = int(np.prod(space2shape(state_space))) if self.store_global_state else 0
        N, f32 = self.n_agents, torch.float32
        act_dim = int(np.prod(self.act_shape)) if self.act_shape else 1
        specs = {"obs": ((N * self.obs_dim,), f32), "actions": ((N * act_dim,), f32), "obs_next": ((N * self.obs_dim,), f32),
                 "rewards": ((N,), f32), "terminals": ((N,), f32), "agent_mask": ((N,), f32)}
        if self.store_global_state:
            specs.update(state=((self.state_dim,), f32), state_next=((self.state_dim,), f32))
        if self.use_actions_mask:
            shp = kwargs["avail_actions_shape"]
            self.n_actions = int(np.prod(shp[k0] if isinstance(shp, dict) else shp))
            specs.update(avail_actions=((N * self.n_actions,), f32), avail_actions_next=((N * self.n_actions,), f32))
        self.specs = specs
        self.soa = _SoA(self.n_size, n_envs, specs, device)
        self.stager = _Stager(n_envs, specs, device)
        self.ptr, self.size = 0, 0
        self.size_dev = torch.zeros(1, dtype=torch.int32, device=device)   # `size` for sampling kernels inside captured graphs

    @property
    def full(self):
        return self.size >= self.n_size

    def clear(self):
        self.ptr, self.size = 0, 0
        self.size_dev.zero_()
        self.soa.zero()

    def _stack(self, v):
        """field value: dict agent -> [n_envs, ...]  or stacked [n_envs, N, ...] / [n_envs, dim]."""
        if isinstance(v, dict):
            parts = [torch.as_tensor(np.asarray(v[k]) if not isinstance(v[k], torch.Tensor) else v[k]).to(torch.float32)
                     .reshape(self.n_envs, -1) for k in self.agent_keys]
            return torch.cat(parts, dim=1)
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
        return t.to(torch.float32).reshape(self.n_envs, -1)

    def store(self, **step_data):                             # memory_tools_marl.py:731-740
        items = {k: self._stack(v) for k, v in step_data.items() if k in self.specs}
        step = self.stager.put(items)
        f = self.soa
        ops.soa_store_step([(f.fields[k], step[k], f.row_bytes[k]) for k in step], self.n_envs, self.ptr)
        self.ptr = (self.ptr + 1) % self.n_size
        if self.size < self.n_size:
            self.size += 1
            self.size_dev.fill_(self.size)

    def gather_into(self, idx, dst):
        """dst: field name -> device tensor [bs, row] (e.g. a learner's staging views); one launch, no host work."""
        f = self.soa
        ops.soa_gather([(dst[k], f.fields[k], f.row_bytes[k]) for k in dst], idx, self.n_envs, self.n_size)

    def sample_indices(self, batch_size=None):                # memory_tools_marl.py:753-754
        bs = self.batch_size if batch_size is None else batch_size
        env_choices = np.random.choice(self.n_envs, bs)
        step_choices = np.random.choice(self.size, bs)
        return env_choices * self.n_size + step_choices

    def sample(self, batch_size=None, indexes=None):          # memory_tools_marl.py:742-765
        assert self.size > 0, "Not enough transitions for off-policy buffer to random sample."
        idx = self.sample_indices(batch_size) if indexes is None else indexes
        idx = torch.as_tensor(np.asarray(idx) if not isinstance(idx, torch.Tensor) else idx)
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        bs, N, f = idx.numel(), self.n_agents, self.soa
        out = {k: torch.empty((bs,) + tuple(shape), device=self.device) for k, (shape, _) in self.specs.items()}
        ops.soa_gather([(out[k], f.fields[k], f.row_bytes[k]) for k in out], idx, self.n_envs, self.n_size)
        sample = {}
        for k, v in out.items():
            if k in ("state", "state_next"):
                sample[k] = v
            else:
                vv = v.view(bs, N, -1)
                sample[k] = {a: (vv[:, i, 0] if vv.shape[-1] == 1 and k != "obs" and k != "obs_next" else vv[:, i])
                             for i, a in enumerate(self.agent_keys)}
        sample["batch_size"] = bs
        return sample

    def finish_path(self, *args, **kwargs):                   # memory_tools_marl.py:766-767
        return

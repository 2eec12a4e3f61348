"""HBM-resident multi-agent replay buffer with the MARL_OffPolicyBuffer surface
(xuance/common/memory_tools_marl.py:634-767): ``store(**step_data)`` takes the reference's nested dicts
(field -> agent -> array [n_envs, ...]) or already stacked tensors, ``sample(batch_size)`` draws uniform
(env, step) pairs with the same two NumPy global-RNG calls and returns the nested per-agent format (views of the
stacked device tensors), so both QMIX_Learner implementations consume it.

Layout: the reference keeps one NumPy array per agent and field, [n_envs][n_size][dim]; under parameter sharing all
agents are homogeneous, so here each field is ONE time-major device array [n_size][n_envs][n_agents*dim]: the per-step
write is one contiguous copy per field and a sampled transition is one contiguous row per field.
HipMARLOffPolicyBufferRNN below is the recurrent / episode variant (MARL_OffPolicyBuffer_RNN, :770-996).
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
        grow = self.size < self.n_size                         # (the filled-slot count for device-side sampling rides in the launch)
        ops.soa_store_step([(f.fields[k], step[k], f.row_bytes[k]) for k in step], self.n_envs, self.ptr,
                           size_dev=self.size_dev if grow else None, new_size=self.size + 1)
        self.ptr = (self.ptr + 1) % self.n_size
        if grow:
            self.size += 1

    def gather_into(self, idx, dst):
        """dst: field name -> device tensor [bs, row] (e.g. a learner's staging views); one launch, no host work."""
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


class HipMARLOffPolicyBufferRNN:
    """MARL_OffPolicyBuffer_RNN (memory_tools_marl.py:770-996) in HBM: per-env staging rows (`episode_data`) collect the
    running episodes, a finished episode is copied as one row into the ring (`data`), `sample` draws whole episodes.

    Every field is [episode][slot][all agents of the step]: obs [T+1][N*obs], actions / rewards / terminals /
    agent_mask [T][N], avail_actions [T+1][N*A], state [T+1][S], filled [T][1] (all float32; the reference keeps bools
    for three of them and converts to float in build_training_data, marl_learner.py:354-385).  `store`,
    `finish_path(s)`, `clear_episodes` and `sample` keep the reference's meaning, including that store_episodes copies
    the WHOLE staging row -- slots past the end of a short episode still hold what an earlier, longer episode of the same
    run_episodes() call left there (:935-949; masked by `filled`).  ptr and size live on the device (`ptr_size`), so
    a step never needs the host to know which envs finished."""

    def __init__(self, agent_keys, state_space=None, obs_space=None, act_space=None, n_envs=1, buffer_size=1,
                 batch_size=1, max_episode_steps=1, device="cuda", **kwargs):
        self.agent_keys = list(agent_keys)
        self.n_agents = N = len(self.agent_keys)
        self.n_envs, self.buffer_size, self.batch_size = n_envs, buffer_size, batch_size
        self.max_eps_len = T = int(max_episode_steps)
        self.device = device
        self.store_global_state = state_space is not None
        self.use_actions_mask = kwargs.get("use_actions_mask", False)
        k0 = self.agent_keys[0]
        self.obs_dim = int(np.prod(space2shape(obs_space[k0])))
        self.state_dim = int(np.prod(space2shape(state_space))) if self.store_global_state else 0
        # field -> (floats per slot, slots)
        self.layout = {"obs": (N * self.obs_dim, T + 1), "actions": (N, T), "rewards": (N, T), "terminals": (N, T),
                       "agent_mask": (N, T), "filled": (1, T)}
        if self.store_global_state:
            self.layout["state"] = (self.state_dim, T + 1)
        if self.use_actions_mask:
            shp = kwargs["avail_actions_shape"]
            self.n_actions = int(np.prod(shp[k0] if isinstance(shp, dict) else shp))
            self.layout["avail_actions"] = (N * self.n_actions, T + 1)
        self.data_keys = list(self.layout)
        mk = lambda rows: {k: torch.zeros(rows, sl, w, device=device) for k, (w, sl) in self.layout.items()}
        self.data = mk(buffer_size)
        # the staging rows of all fields are views of ONE allocation: clear_episodes() is one launch
        sizes = {k: n_envs * sl * w for k, (w, sl) in self.layout.items()}
        self._staging = torch.zeros(sum((v + 3) // 4 * 4 for v in sizes.values()), device=device)
        self.episode_data, off = {}, 0
        for k, (w, sl) in self.layout.items():
            self.episode_data[k] = self._staging[off:off + sizes[k]].view(n_envs, sl, w)
            off += (sizes[k] + 3) // 4 * 4
        self.ptr_size = torch.zeros(2, dtype=torch.int32, device=device)       # ptr, size
        self.size_dev = self.ptr_size[1:2]
        self._ones = torch.ones(n_envs, 1, device=device)
        self.stager = _Stager(n_envs, {k: ((w,), torch.float32) for k, (w, _) in self.layout.items() if k != "filled"}, device)
        self._steps = torch.zeros(n_envs, dtype=torch.int32, device=device)

    # -- host views of the device counters (one sync each; only the reference-style host API uses them) ---------------
    @property
    def ptr(self):
        return int(self.ptr_size[0].item())

    @property
    def size(self):
        return int(self.ptr_size[1].item())

    @property
    def full(self):
        return self.size >= self.buffer_size

    def clear(self):                                           # :822-857
        for v in self.data.values():
            v.zero_()
        self.ptr_size.zero_()

    def clear_episodes(self):                                  # :859-902
        self._staging.zero_()

    def _stack(self, v, width):
        if isinstance(v, dict):
            parts = [torch.as_tensor(np.asarray(v[k]) if not isinstance(v[k], torch.Tensor) else v[k]).to(torch.float32)
                     .reshape(self.n_envs, -1) for k in self.agent_keys]
            return torch.cat(parts, dim=1)
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
        return t.to(torch.float32).reshape(self.n_envs, width)

    def _dev_steps(self, steps):
        if isinstance(steps, torch.Tensor) and steps.is_cuda:
            return steps.to(torch.int32).contiguous()
        self._steps.copy_(torch.as_tensor(np.asarray(steps), dtype=torch.int32))
        return self._steps

    def store(self, **step_data):                              # :904-921
        steps = self._dev_steps(step_data["episode_steps"])
        items = {k: self._stack(v, self.layout[k][0]) for k, v in step_data.items() if k in self.layout and k != "filled"}
        dev = self.stager.put(items)
        fields = [(self.episode_data[k], dev[k], None, 4 * self.layout[k][0], self.layout[k][1], 0) for k in dev]
        fields.append((self.episode_data["filled"], self._ones, None, 4, self.max_eps_len, 0))
        ops.episode_store_step(fields, steps, self.n_envs)

    def finish_paths(self, done, end_step, obs=None, state=None, avail_actions=None, gate=None, advance=True):
        """finish_path (:951-968) for every env with done != 0, in env order, in two launches.  done [n_envs] f32,
        end_step [n_envs] int32 (info['episode_step']), terminal obs / state / avail_actions as in `store`."""
        term = {"obs": obs, "state": state if self.store_global_state else None,
                "avail_actions": avail_actions if self.use_actions_mask else None}
        term = {k: self._stack(v, self.layout[k][0]).contiguous() for k, v in term.items() if v is not None}
        self._term_keep = term                                  # keep the temporaries alive until the launch ran
        fields = [(self.data[k], self.episode_data[k], term.get(k), 4 * w, sl, 1 if k == "filled" else 0)
                  for k, (w, sl) in self.layout.items()]
        ops.episode_finish(fields, done.to(torch.float32).contiguous(), self._dev_steps(end_step), self.ptr_size,
                           self.n_envs, self.buffer_size, gate=gate, advance=advance)   # gate: device scalar, 0 = close nothing

    def store_and_finish(self, step_data, episode_steps, done, end_step, obs=None, state=None, avail_actions=None, gate=None,
                         loop_gate=None):
        """store(**step_data, episode_steps=...) followed by finish_paths(done, end_step, ..., gate=gate, advance=False) as ONE
        launch (device tensors only: the captured vector step of the agents); the ring's {ptr, size} are left to the caller
        (xrl_marl_loop_gate advances them).  loop_gate: the keyword arguments of ops.marl_loop_gate -- the gate then rides in the
        same launch (xrl_episode_store_finish_gate: the last block to finish carries it)."""
        steps = self._dev_steps(episode_steps)
        items = {k: self._stack(v, self.layout[k][0]) for k, v in step_data.items() if k in self.layout and k != "filled"}
        items["filled"] = self._ones
        term = {"obs": obs, "state": state if self.store_global_state else None,
                "avail_actions": avail_actions if self.use_actions_mask else None}
        term = {k: self._stack(v, self.layout[k][0]).contiguous() for k, v in term.items() if v is not None}
        self._term_keep = (term, items)                         # keep the temporaries alive until the launch ran
        fields = [(self.data[k], self.episode_data[k], term.get(k), 4 * w, sl, 1 if k == "filled" else 0, items.get(k))
                  for k, (w, sl) in self.layout.items()]
        if loop_gate is not None:
            if getattr(self, "_ticket", None) is None:
                self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
            ops.episode_store_finish_gate(fields, steps, done.to(torch.float32).contiguous(), self._dev_steps(end_step), self.ptr_size,
                                          self.n_envs, self.buffer_size, gate, loop_gate, self._ticket)
            return
        ops.episode_store_finish(fields, steps, done.to(torch.float32).contiguous(), self._dev_steps(end_step), self.ptr_size,
                                 self.n_envs, self.buffer_size, gate=gate)

    def finish_path(self, i_env, **terminal_data):            # :951-968, one env (the reference's call)
        done = torch.zeros(self.n_envs, device=self.device)
        done[i_env] = 1.0
        end = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        end[i_env] = int(terminal_data["episode_step"])

        def full(v, k):                                        # the reference passes one env's data: place it in row i_env
            if v is None:
                return None
            row = torch.cat([torch.as_tensor(np.asarray(v[a])).to(torch.float32).reshape(-1) for a in self.agent_keys]) \
                if isinstance(v, dict) else torch.as_tensor(np.asarray(v)).to(torch.float32).reshape(-1)
            out = torch.zeros(self.n_envs, self.layout[k][0], device=self.device)
            out[i_env] = row.to(self.device)
            return out
        self.finish_paths(done, end, obs=full(terminal_data.get("obs"), "obs"), state=full(terminal_data.get("state"), "state"),
                          avail_actions=full(terminal_data.get("avail_actions"), "avail_actions"))

    def gather_into(self, idx, dst):
        """dst: field -> time-major device tensor [slots][B][row] (a learner's staging tensors); one launch."""
        ops.episode_gather([(dst[k], self.data[k], None, 4 * self.layout[k][0], self.layout[k][1], 0) for k in dst],
                           idx, idx.numel())

    def draw_into(self, idx_out, dst, seed, counter, counter_dev):
        """Uniform draw of idx_out.numel() episodes (xrl_sample_replay_indices' stream, following the filling ring through size_dev)
        + gather_into, as ONE launch."""
        ops.episode_gather_sampled([(dst[k], self.data[k], None, 4 * self.layout[k][0], self.layout[k][1], 0) for k in dst],
                                   idx_out, idx_out.numel(), self.buffer_size, self.size_dev, seed, counter, counter_dev)

    def sample(self, batch_size=None, indexes=None):           # :970-996
        size = self.size
        assert size > 0, "You need to first store experience data into the buffer!"
        bs = self.batch_size if batch_size is None else batch_size
        idx = np.random.choice(size, bs) if indexes is None else np.asarray(indexes)
        idx = torch.as_tensor(idx).to(device=self.device, dtype=torch.int64).contiguous()
        N, out = self.n_agents, {}
        tm = {k: torch.empty(sl, bs, w, device=self.device) for k, (w, sl) in self.layout.items()}
        self.gather_into(idx, tm)
        for k, v in tm.items():
            v = v.transpose(0, 1)                               # [B][slots][row] views, the reference's axis order
            if k == "filled":
                out[k] = v[..., 0]
            elif k == "state":
                out[k] = v
            else:
                vv = v.reshape(bs, v.shape[1], N, -1)
                out[k] = {a: (vv[:, :, i, 0] if k in ("actions", "rewards", "terminals", "agent_mask") else vv[:, :, i])
                          for i, a in enumerate(self.agent_keys)}
        out["batch_size"], out["sequence_length"] = bs, self.max_eps_len
        return out

"""Multi-agent subprocess vector env over shared, page-locked host memory -- the multi-agent twin of shm_vec_env.py
(SURVEY.md section 8f.3).

The reference's SubprocVecMultiAgentEnv (xuance/environment/vector_envs/subprocess/subproc_vec_maenv.py:8-170) pickles, per
worker and per step, a list of (obs dict, reward dict, terminated dict, truncated, info dict incl. state / avail_actions /
reset_*) through a Pipe.  Here every worker writes its envs' slice of ONE shared block of fixed-size arrays -- actions in;
per-agent observations, global state, action-availability masks (each also in its post-reset form), rewards, terminated
flags, agent masks, truncated, episode step / per-agent episode scores out -- and the pipes carry a one-byte command and a
one-byte acknowledgement.  The block is registered with the HIP runtime, so ``step_to_device`` moves a whole vector step
into HBM with one asynchronous DMA, already in the [n_envs, n_agents, ...] arrays the device-side agents consume.

Same surface and auto-reset contract as the reference class (and as DummyVecMultiAgentEnv): ``reset() -> (obs_list,
infos)``; ``step_async(actions)`` with actions a list of {agent: action} dicts or an [n_envs, n_agents] array;
``step_wait() -> (obs_list, reward dicts, terminated dicts, truncated list, infos)`` where infos[e] carries ``state``,
``avail_actions``, ``agent_mask``, ``episode_step``, ``episode_score`` and, when the episode ended (all agents terminated, or
truncated: worker step_env, :9-16), ``reset_obs`` / ``reset_avail_actions`` / ``reset_state``; ``buf_obs / buf_state /
buf_avail_actions``; ``agents / num_agents / state_space / max_episode_steps``; ``in_series`` envs per process,
``env_seed + index`` seeding (:19-22, 77-84); ``close()``."""
import multiprocessing as mp

import numpy as np
import torch

from ..spaces import space2shape

_F32, _U8, _I32 = np.float32, np.uint8, np.int32


def _fields(n, N, O, S, A):
    return [("actions", _F32, (n, N)), ("obs", _F32, (n, N, O)), ("state", _F32, (n, S)), ("avail", _F32, (n, N, A)),
            ("reset_obs", _F32, (n, N, O)), ("reset_state", _F32, (n, S)), ("reset_avail", _F32, (n, N, A)),
            ("rewards", _F32, (n, N)), ("terminated", _U8, (n, N)), ("agent_mask", _F32, (n, N)), ("truncated", _U8, (n,)),
            ("episode_step", _I32, (n,)), ("episode_score", _F32, (n, N))]


def _layout(fields):
    off, lay = 0, {}
    for name, dt, shape in fields:
        nb = int(np.prod(shape)) * np.dtype(dt).itemsize
        lay[name] = (off, nb, dt, shape)
        off = (off + nb + 63) // 64 * 64
    return lay, off


def _views(block, lay):
    raw = block.numpy()
    return {k: raw[o:o + nb].view(dt).reshape(shape) for k, (o, nb, dt, shape) in lay.items()}


def _put(dst, per_agent, agents):
    for i, a in enumerate(agents):
        dst[i] = per_agent[a]


def _worker(remote, parent_remote, env_fns, env_seed, lo, block, lay, agents):
    parent_remote.close()
    envs = [fn() if env_seed is None else fn(env_seed=env_seed + i) for i, fn in enumerate(env_fns)]   # :19-22
    v = _views(block, lay)

    def publish(e, obs, info, prefix=""):
        _put(v[prefix + "obs"][e], obs, agents)
        v[prefix + "state"][e] = info["state"]
        _put(v[prefix + "avail"][e], info["avail_actions"], agents)
    try:
        while True:
            cmd = remote.recv_bytes()
            if cmd == b"s":
                for i, env in enumerate(envs):
                    e = lo + i
                    act = {a: int(v["actions"][e, j]) for j, a in enumerate(agents)}
                    obs, rew, term, trunc, info = env.step(act)
                    publish(e, obs, info)
                    _put(v["rewards"][e], rew, agents)
                    _put(v["terminated"][e], term, agents)
                    v["truncated"][e] = trunc
                    mask = info.get("agent_mask")
                    v["agent_mask"][e] = 1.0 if mask is None else [float(mask[a]) for a in agents]
                    v["episode_step"][e] = info.get("episode_step", 0)
                    score = info.get("episode_score")
                    v["episode_score"][e] = 0.0 if score is None else [float(score[a]) for a in agents]
                    if all(term.values()) or trunc:                     # step_env, :10-15
                        obs_r, info_r = env.reset()
                        publish(e, obs_r, info_r, "reset_")
                remote.send_bytes(b"k")
            elif cmd == b"r":
                for i, env in enumerate(envs):
                    obs, info = env.reset()
                    publish(lo + i, obs, info)
                    v["agent_mask"][lo + i] = 1.0
                    v["episode_step"][lo + i] = 0
                    v["episode_score"][lo + i] = 0.0
                remote.send_bytes(b"k")
            elif cmd == b"c":
                break
    except Exception as ex:                  # tell the parent instead of dying silently: the forked siblings hold copies of
        import traceback                     # this pipe's ends, so the parent would never see EOF and wait for ever
        remote.send_bytes(b"x" + traceback.format_exc().encode()[-2000:])
    finally:
        for env in envs:
            env.close()
        remote.close()


class ShmSubprocVecMultiAgentEnv:
    def __init__(self, env_fns, env_seed=None, in_series=1, device="cuda"):
        self.waiting, self.closed = False, False
        self.num_envs = n = len(env_fns)
        assert n % in_series == 0, "Number of envs must be divisible by number of envs to run in series"
        probe = env_fns[0]() if env_seed is None else env_fns[0](env_seed=env_seed)
        self.agents = self.agent_keys = list(probe.agents)
        self.num_agents = N = len(self.agents)
        self.observation_space, self.action_space, self.state_space = probe.observation_space, probe.action_space, probe.state_space
        self.max_episode_steps = probe.max_episode_steps
        probe.close()
        k0 = self.agents[0]
        O, A, S = int(space2shape(self.observation_space[k0])[0]), int(self.action_space[k0].n), int(space2shape(self.state_space)[0])
        self.lay, nbytes = _layout(_fields(n, N, O, S, A))
        self.block = torch.zeros(nbytes, dtype=torch.uint8).share_memory_()
        self.v = _views(self.block, self.lay)
        # fork() BEFORE the first HIP call of this object (see shm_vec_env.py)
        ctx = mp.get_context("fork")
        self.n_remotes = n // in_series
        bounds = np.array_split(np.arange(n), self.n_remotes)
        self.remotes, work = zip(*[ctx.Pipe() for _ in range(self.n_remotes)])
        self.ps = []
        for remote, wr, idx in zip(self.remotes, work, bounds):
            seed = None if env_seed is None else env_seed + int(idx[0])                 # :77-84
            p = ctx.Process(target=_worker, args=(wr, remote, [env_fns[i] for i in idx], seed, int(idx[0]), self.block, self.lay,
                                                  self.agents), daemon=True)
            p.start()
            self.ps.append(p)
        for wr in work:
            wr.close()
        self.device, self._pinned, self._dma_done = device, False, None
        if torch.cuda.is_available() and str(device).startswith("cuda"):
            err = torch.cuda.cudart().cudaHostRegister(self.block.data_ptr(), nbytes, 0)     # page-lock in place
            self._pinned = int(err) == 0
            self.dev_block = torch.zeros(nbytes, dtype=torch.uint8, device=device)
            tdt = {_F32: torch.float32, _U8: torch.uint8, _I32: torch.int32}
            self.dev = {k: self.dev_block[o:o + nb].view(tdt[dt]).view(shape) for k, (o, nb, dt, shape) in self.lay.items()
                        if k != "actions"}
        self.buf_obs = [{} for _ in range(n)]
        self.buf_state = [np.zeros(S, np.float32) for _ in range(n)]
        self.buf_avail_actions = [{} for _ in range(n)]
        self.buf_info = [{} for _ in range(n)]

    # -- reference surface ---------------------------------------------------------------------------------------------
    def _per_agent(self, arr):
        return {a: arr[i].copy() for i, a in enumerate(self.agents)}

    def _refresh(self):
        v = self.v
        for e in range(self.num_envs):
            self.buf_obs[e] = self._per_agent(v["obs"][e])
            self.buf_state[e] = v["state"][e].copy()
            self.buf_avail_actions[e] = self._per_agent(v["avail"][e])

    @staticmethod
    def _ack(r):
        msg = r.recv_bytes()
        if msg != b"k":
            raise RuntimeError("vector-env worker failed:\n" + msg[1:].decode(errors="replace"))

    def _all(self, cmd):
        for r in self.remotes:
            r.send_bytes(cmd)
        for r in self.remotes:
            self._ack(r)

    def reset(self):
        self._assert_not_closed()
        self._all(b"r")
        self._refresh()
        self.buf_info = [{"state": self.buf_state[e], "avail_actions": self.buf_avail_actions[e],
                          "agent_mask": {a: True for a in self.agents}, "episode_step": 0,
                          "episode_score": {a: 0.0 for a in self.agents}} for e in range(self.num_envs)]
        return list(self.buf_obs), list(self.buf_info)

    def step_async(self, actions):
        self._assert_not_closed()
        if self._dma_done is not None:                              # the previous step_to_device copy still reads the block
            self._dma_done.synchronize()
            self._dma_done = None
        if isinstance(actions, (list, tuple)) and isinstance(actions[0], dict):
            actions = [[d[a] for a in self.agents] for d in actions]
        self.v["actions"][...] = np.asarray(actions, np.float32).reshape(self.num_envs, self.num_agents)
        for r in self.remotes:
            r.send_bytes(b"s")
        self.waiting = True

    def _wait(self):
        self.waiting = False
        for r in self.remotes:
            self._ack(r)

    def step_wait(self):
        self._assert_not_closed()
        self._wait()
        v = self.v
        self._refresh()
        rew, term, trunc, infos = [], [], [], []
        for e in range(self.num_envs):
            t = {a: bool(v["terminated"][e, i]) for i, a in enumerate(self.agents)}
            tr = bool(v["truncated"][e])
            info = {"state": self.buf_state[e], "avail_actions": self.buf_avail_actions[e],
                    "agent_mask": {a: bool(v["agent_mask"][e, i]) for i, a in enumerate(self.agents)},
                    "episode_step": int(v["episode_step"][e]),
                    "episode_score": {a: float(v["episode_score"][e, i]) for i, a in enumerate(self.agents)}}
            if all(t.values()) or tr:
                info["reset_obs"] = self._per_agent(v["reset_obs"][e])
                info["reset_avail_actions"] = self._per_agent(v["reset_avail"][e])
                info["reset_state"] = v["reset_state"][e].copy()
            rew.append({a: float(v["rewards"][e, i]) for i, a in enumerate(self.agents)})
            term.append(t); trunc.append(tr); infos.append(info)
        self.buf_info = infos
        return list(self.buf_obs), rew, term, trunc, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # -- device path ---------------------------------------------------------------------------------------------------
    def step_to_device(self, actions):
        """One vector step whose results land in HBM with a single asynchronous copy of the shared block; returns the
        device tensors by field name (obs [n, N, O], state [n, S], avail [n, N, A], their reset_ forms, rewards, terminated,
        agent_mask [n, N], truncated, episode_step [n], episode_score [n, N]).  actions: [n, N] device tensor or host data."""
        if isinstance(actions, torch.Tensor):
            actions = actions.detach().to("cpu", torch.float32).numpy()
        self.step_async(actions)
        self._wait()
        self.dev_block.copy_(self.block, non_blocking=self._pinned)
        if self._pinned:
            self._dma_done = torch.cuda.Event()
            self._dma_done.record()
        return self.dev

    def close(self):
        if self.closed:
            return
        if self.waiting:
            self._wait()
        for r in self.remotes:
            r.send_bytes(b"c")
        for p in self.ps:
            p.join(timeout=5)
        if self._pinned:
            torch.cuda.cudart().cudaHostUnregister(self.block.data_ptr())
        self.closed = True

    def _assert_not_closed(self):
        assert not self.closed, "Trying to operate on a ShmSubprocVecMultiAgentEnv after calling close()"

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

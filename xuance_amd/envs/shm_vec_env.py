"""Subprocess vector env whose step data travels through shared, page-locked host memory.

The reference's SubprocVecEnv (xuance/environment/vector_envs/subprocess/subproc_vec_env.py:8-150) pickles every
observation / reward / flag / info dict through a Pipe per worker per step, and the agent then copies the stacked arrays
once more on their way to the device.  Here the workers write their slice of ONE shared block -- actions in; observation,
post-reset observation, reward, terminated, truncated, episode step / score out -- and the pipes carry a one-byte command
and a one-byte acknowledgement.  The block is registered with the HIP runtime (hipHostRegister), so ``step_to_device``
moves a whole vector step into HBM with a single asynchronous DMA; the device tensors it returns are what
``HipOnPolicyBuffer.store`` / ``xrl_soa_store_step`` consume.

Same surface and auto-reset contract as the reference class: ``reset() -> (obs, infos)``, ``step_async(actions)``,
``step_wait() -> (obs, rewards, terminated, truncated, infos)`` with ``infos[i]["reset_obs"]`` when an episode ended
(worker step_env, :9-14), ``close()``, ``num_envs / observation_space / action_space / buf_obs / max_episode_steps``,
``in_series`` envs per process, ``env_seed + index`` seeding (:19-22, 68-73).  Of an env's ``info`` only
``episode_step`` and ``episode_score`` are transported (fixed-size fields); other keys stay in the worker."""
import multiprocessing as mp

import numpy as np
import torch

from ..spaces import space2shape


def _layout(n, obs_shape, obs_dtype, act_dim):
    """Byte offsets of the per-field arrays inside the shared block (every field 64-byte aligned)."""
    obs_n = int(np.prod(obs_shape)) * np.dtype(obs_dtype).itemsize
    fields = [("actions", n * act_dim * 4), ("obs", n * obs_n), ("reset_obs", n * obs_n), ("rewards", n * 4),
              ("terminated", n), ("truncated", n), ("episode_step", n * 4), ("episode_score", n * 4)]
    off, out = 0, {}
    for name, nb in fields:
        out[name] = (off, nb)
        off = (off + nb + 63) // 64 * 64
    return out, off


def _views(block, lay, n, obs_shape, obs_dtype, act_dim):
    """NumPy views of the shared block (block: uint8 torch tensor in shared memory)."""
    raw = block.numpy()
    v = lambda name, dt, shape: raw[lay[name][0]:lay[name][0] + lay[name][1]].view(dt).reshape(shape)
    return dict(actions=v("actions", np.float32, (n, act_dim)), obs=v("obs", obs_dtype, (n,) + tuple(obs_shape)),
                reset_obs=v("reset_obs", obs_dtype, (n,) + tuple(obs_shape)), rewards=v("rewards", np.float32, (n,)),
                terminated=v("terminated", np.uint8, (n,)), truncated=v("truncated", np.uint8, (n,)),
                episode_step=v("episode_step", np.int32, (n,)), episode_score=v("episode_score", np.float32, (n,)))


def _worker(remote, parent_remote, env_fns, env_seed, lo, block, lay, n, obs_shape, obs_dtype, act_dim, discrete):
    parent_remote.close()
    envs = [fn() if env_seed is None else fn(env_seed=env_seed + i) for i, fn in enumerate(env_fns)]   # :19-22
    v = _views(block, lay, n, obs_shape, obs_dtype, act_dim)
    try:
        while True:
            cmd = remote.recv_bytes()
            if cmd == b"s":
                for i, env in enumerate(envs):
                    e = lo + i
                    a = v["actions"][e]
                    obs, rew, term, trunc, info = env.step(int(a[0]) if discrete else a.copy())
                    v["obs"][e], v["rewards"][e], v["terminated"][e], v["truncated"][e] = obs, rew, term, trunc
                    v["episode_step"][e] = info.get("episode_step", 0)
                    v["episode_score"][e] = info.get("episode_score", 0.0)
                    if term or trunc:                                   # step_env, :10-13
                        v["reset_obs"][e] = env.reset()[0]
                remote.send_bytes(b"k")
            elif cmd == b"r":
                for i, env in enumerate(envs):
                    v["obs"][lo + i] = env.reset()[0]
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


class ShmSubprocVecEnv:
    def __init__(self, env_fns, env_seed=None, in_series=1, device="cuda"):
        self.waiting, self.closed = False, False
        self.num_envs = n = len(env_fns)
        probe = env_fns[0]() if env_seed is None else env_fns[0](env_seed=env_seed)
        self.observation_space, self.action_space = probe.observation_space, probe.action_space
        self.max_episode_steps = getattr(probe, "max_episode_steps", None)
        probe.close()
        self.obs_shape = tuple(space2shape(self.observation_space))
        self.obs_dtype = np.dtype(getattr(self.observation_space, "dtype", np.float32) or np.float32)
        self.discrete = hasattr(self.action_space, "n")
        self.act_dim = 1 if self.discrete else int(np.prod(space2shape(self.action_space)))
        self.lay, nbytes = _layout(n, self.obs_shape, self.obs_dtype, self.act_dim)
        self.block = torch.zeros(nbytes, dtype=torch.uint8).share_memory_()
        self.v = _views(self.block, self.lay, n, self.obs_shape, self.obs_dtype, self.act_dim)
        # the worker processes are fork()ed BEFORE this object makes its first HIP call (hipHostRegister, the device
        # mirror): a child forked from a process whose HIP runtime is already live inherits runtime state it must never
        # use.  (If the caller initialised HIP earlier, create the vector env first -- or the workers simply never touch it:
        # they only see NumPy views of the shared block.)
        ctx = mp.get_context("fork")
        self.n_remotes = n // in_series
        bounds = np.array_split(np.arange(n), self.n_remotes)
        self.remotes, work = zip(*[ctx.Pipe() for _ in range(self.n_remotes)])
        self.ps = []
        for r, (remote, wr, idx) in enumerate(zip(self.remotes, work, bounds)):
            seed = None if env_seed is None else env_seed + int(idx[0])                 # :68-73
            p = ctx.Process(target=_worker, args=(wr, remote, [env_fns[i] for i in idx], seed, int(idx[0]), self.block, self.lay,
                                                  n, self.obs_shape, self.obs_dtype, self.act_dim, self.discrete), daemon=True)
            p.start()
            self.ps.append(p)
        for wr in work:
            wr.close()
        self.device = device
        self._pinned = False
        self._dma_done = None                                       # event of the last step_to_device copy
        if torch.cuda.is_available() and str(device).startswith("cuda"):
            # page-lock the shared block in place: H2D copies from it are true asynchronous DMAs
            err = torch.cuda.cudart().cudaHostRegister(self.block.data_ptr(), nbytes, 0)
            self._pinned = int(err) == 0
            self.dev_block = torch.zeros(nbytes, dtype=torch.uint8, device=device)
            d = lambda name, dt, shape: self.dev_block[self.lay[name][0]:self.lay[name][0] + self.lay[name][1]].view(dt).view(shape)
            tdt = torch.uint8 if self.obs_dtype == np.uint8 else torch.float32
            self.dev = dict(obs=d("obs", tdt, (n,) + self.obs_shape), reset_obs=d("reset_obs", tdt, (n,) + self.obs_shape),
                            rewards=d("rewards", torch.float32, (n,)), terminated=d("terminated", torch.uint8, (n,)),
                            truncated=d("truncated", torch.uint8, (n,)), episode_step=d("episode_step", torch.int32, (n,)),
                            episode_score=d("episode_score", torch.float32, (n,)))
        self.buf_obs = np.zeros((n,) + self.obs_shape, self.obs_dtype)

    # -- reference surface ---------------------------------------------------------------------------------------------
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
        self.buf_obs = self.v["obs"].copy()
        return self.buf_obs.copy(), [{} for _ in range(self.num_envs)]

    def step_async(self, actions):
        self._assert_not_closed()
        if self._dma_done is not None:                              # the previous step_to_device copy still reads the block
            self._dma_done.synchronize()
            self._dma_done = None
        self.v["actions"][...] = np.asarray(actions, np.float32).reshape(self.num_envs, self.act_dim)
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
        term, trunc = v["terminated"].astype(bool), v["truncated"].astype(bool)
        infos = []
        for i in range(self.num_envs):
            info = {"episode_step": int(v["episode_step"][i]), "episode_score": float(v["episode_score"][i])}
            if term[i] or trunc[i]:
                info["reset_obs"] = v["reset_obs"][i].copy()
            infos.append(info)
        self.buf_obs = v["obs"].copy()
        return self.buf_obs.copy(), v["rewards"].copy(), term, trunc, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # -- device path ---------------------------------------------------------------------------------------------------
    def step_to_device(self, actions):
        """One vector step whose results land in HBM with a single asynchronous copy of the shared block; returns the
        device tensors (obs, reset_obs, rewards, terminated, truncated, episode_step, episode_score).  `actions` may be
        a device tensor (copied into the block) or host data."""
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
        assert not self.closed, "Trying to operate on a ShmSubprocVecEnv after calling close()"

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

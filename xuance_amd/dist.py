"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The path shards over environments: every rank owns n_envs/world envs and its own rollout
buffer, parameters are replicated, and the ONLY data-path collective is one all-reduce (mean) of the flat
gradient buffer per optimiser step -- 136 KB..570 KB, latency-bound on the xGMI mesh, hence a single flat message
instead of DDP's per-bucket calls.  Observation / return statistics stay PER RANK, as in the reference's default
(RunningMeanStd(use_mpi=False), statistic_tools.py:65-110); only its `use_tensor_memory` variant all-reduces batch moments
per step (tensor_statistics.py:48-58) -- `allreduce_moments_` below is that rule for callers that want it, the agents do
not call it.
"""
import os

import torch
import torch.distributed as dist


def init_distributed_mode(backend=None):
    """xuance/torch/utils/operations.py:11-28, but the backend falls back to gloo when no GPU is present
    (CPU tests) and the rendezvous address is 127.0.0.1."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


_avg_ok = True


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_mean_(flat):
    """In-place mean over ranks of one flat tensor (what DDP does to gradients, as ONE message)."""
    if not dist.is_initialized():
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("XRL_DIST_STUB") != "1":
            raise RuntimeError("allreduce_mean_: WORLD_SIZE > 1 but torch.distributed is not initialised -- the ranks would "
                               "train independent replicas (call xuance_amd.dist.init_distributed_mode() first)")
        return flat
    if dist.get_world_size() == 1:
        return flat
    global _avg_ok
    if _avg_ok and dist.get_backend() == "nccl":           # RCCL averages inside the collective: no extra launch
        try:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            return flat
        except (RuntimeError, ValueError):                 # a build without ncclAvg: fall back for good
            _avg_ok = False
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)            # gloo (CPU tests) has no AVG
    flat.div_(dist.get_world_size())
    return flat


def allreduce_moments_(batch_mean, batch_var, batch_count):
    """The reference's `_sync_distributed_moments` (xuance/torch/utils/tensor_statistics.py:48-58) as ONE message: the
    per-rank batch means and variances are AVERAGED over the ranks (the between-rank variance is ignored -- its rule, kept),
    the counts are summed.  Tensors of any (equal) shape for mean / var, a scalar tensor for the count; in place."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return batch_mean, batch_var, batch_count
    n = batch_mean.numel()
    # one float64 message: float32 means / variances travel exactly, and so does the count (summed counts above 2^24 would lose
    # their integer part in float32; the reference reduces the count in its own dtype)
    packed = torch.cat([batch_mean.reshape(-1).double(), batch_var.reshape(-1).double(), batch_count.reshape(-1)[:1].double()])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    w = dist.get_world_size()
    batch_mean.copy_((packed[:n] / w).view_as(batch_mean).to(batch_mean.dtype))
    batch_var.copy_((packed[n:2 * n] / w).view_as(batch_var).to(batch_var.dtype))
    batch_count.copy_(packed[2 * n:2 * n + 1].view_as(batch_count).to(batch_count.dtype))
    return batch_mean, batch_var, batch_count


_capturable = None


def captured_allreduce_works(device):
    """Capture an all-reduce of a known tensor into a hipGraph (ops.Graph on a side stream: the process group's own stream
    is forked from and joined to it by the events torch records, so the collective becomes part of the graph), replay it
    twice and compare with the known answer.  True only if every rank saw the right numbers both times."""
    from . import ops
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    try:
        x = torch.full((1024,), float(rank + 1), device=device)
        y = torch.zeros_like(x)
        warm = x.clone()
        dist.all_reduce(warm, op=dist.ReduceOp.SUM)        # communicator set-up happens outside the capture
        torch.cuda.synchronize()
        g = ops.Graph()
        with g:
            torch.mul(x, 1.0, out=y)
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
            torch.mul(y, 1.0 / world, out=y)
        for _ in range(2):
            y.zero_()
            g.launch()
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(y, torch.full_like(y, (world + 1) / 2.0)))
    except Exception:                                      # a build whose RCCL / process group refuses stream capture
        ok = False
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() == 1.0)


def collective_capturable(device):
    """May the gradient all-reduce be captured into the update graph?  RCCL only (gloo reduces on the host), more than one
    rank, not disabled by XRL_DIST_GRAPH_COLLECTIVE=0, and proven once per process by captured_allreduce_works."""
    global _capturable, _avg_ok
    if _capturable is None:
        _capturable = False
        if dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl" and \
                os.environ.get("XRL_DIST_GRAPH_COLLECTIVE", "1") != "0":
            allreduce_mean_(torch.ones(4, device=device))  # settles whether ncclAvg exists before anything is captured
            _capturable = captured_allreduce_works(device)
    return _capturable


class GradientExchange:
    """Exchange buffers of xrl_reduce_adam_exchange for gradients of up to P values: this rank's buffer (fine-grained
    device memory) and every peer's, mapped through IPC handles that travel over the process group once.  After this
    set-up the data path of an update makes no collective call at all (include/xrl_hip.h, csrc/optim.hip)."""

    def __init__(self, P, device, max_spins=None):
        from . import ops
        from ._lib import Exchange
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert 1 < self.world <= ops.XC_MAX_RANKS and (P + 255) // 256 <= ops.XC_MAX_GROUPS
        self.stride4 = (P + 255) // 256 * 64
        self.nbytes = ops.xc_bytes(self.stride4)
        self.ptr, self.peers, handle, err = None, None, None, None
        # every rank makes the same collective calls whatever fails locally (a rank that raised early would leave its
        # peers waiting in a collective it never joins)
        try:
            self.ptr, handle = ops.ipc_alloc(self.nbytes)
        except Exception as ex:
            err = ex
        handles = [None] * self.world
        dist.all_gather_object(handles, handle)
        if err is None and all(h is not None for h in handles):
            try:
                self.peers = [self.ptr if r == self.rank else ops.ipc_open(h) for r, h in enumerate(handles)]
            except Exception as ex:
                err = ex
        oks = [None] * self.world
        dist.all_gather_object(oks, self.peers is not None)    # (also the barrier: nobody polls a buffer not mapped everywhere)
        if not all(oks):
            self.close()
            raise RuntimeError(f"GradientExchange: set-up failed on rank(s) {[r for r, o in enumerate(oks) if not o]}: {err}")
        x = Exchange()
        for r, p in enumerate(self.peers):
            x.base[r] = p
        # polling rounds a group waits for its peers' rows before the launch gives up (sleep 2 between rounds: ~0.3 s at the default).
        # XRL_DIST_EXCHANGE_SPINS raises it where ranks may be far apart for reasons other than a dead peer -- several ranks time-sharing
        # ONE GPU (test boxes): a rank can sit in this wait for as long as its peers take to capture their first update graph
        if max_spins is None:
            max_spins = int(os.environ.get("XRL_DIST_EXCHANGE_SPINS", 4_000_000))
        x.stride4, x.world, x.rank, x.max_spins, x.inv_world = self.stride4, self.world, self.rank, int(max_spins), 1.0 / self.world
        self.struct = x

    def clear(self):
        """Forget the step numbers of an earlier life (a checkpoint was loaded: the optimiser step may repeat)."""
        from . import ops
        torch.cuda.synchronize()
        dist.barrier()                                         # every rank's launches that poll these flags are over
        ops.ipc_clear(self.ptr, self.nbytes)
        torch.cuda.synchronize()
        dist.barrier()

    def close(self):
        from . import ops
        torch.cuda.synchronize()
        for r, p in enumerate(self.peers or []):
            if r != self.rank:
                ops.ipc_close(p)
        if self.ptr is not None:
            ops.ipc_free(self.ptr)
        self.peers, self.ptr = None, None


_exchange_ok = None


def exchange_selftest(device, P=3000, rounds=5):
    """xrl_reduce_adam_exchange on a scratch problem (slabs that depend on the rank but are known to every rank, an
    optimiser of its own, both buffer parities and their reuse): in every round every rank must find, bit for bit, (the
    fp32 sum in rank order of the ranks' slab sums) * (1 / world) in the gradient the launch reports, see no time-out, and
    hold the same gradient and parameters as every other rank (compared through the process group)."""
    from . import ops
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        # (raises on every rank or on none; ranks that TIME-SHARE one GPU, as in the tests, need a longer wait)
        xc = GradientExchange(P, device, max_spins=int(os.environ.get("XRL_DIST_SELFTEST_SPINS", 400_000)))
    except RuntimeError:
        return False
    ok = True
    g = torch.Generator(device="cpu").manual_seed(1234)
    slabs_all = torch.randn(rounds, world, 3, P, generator=g)   # the same numbers on every rank
    params = torch.zeros(P, device=device)
    grad, m, v = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
    n_part = (P + 255) // 256
    sumsq = torch.zeros(n_part, dtype=torch.float64, device=device)
    sync = torch.zeros(4 + n_part + 8, dtype=torch.int32, device=device)
    state = ops.adam_state_tensor(1e-3, 1, 1.0, 1e-5, 0.0, device=device)
    for k in range(rounds):
        try:                                                     # local work may fail; the collectives below always run
            ops.reduce_adam(slabs_all[k, rank].to(device).contiguous(), 3, P, params, grad, m, v, P, state, sumsq, 0.0, [], sync,
                            exchange=xc)
            torch.cuda.synchronize()
            ok = ok and int(sync[2]) == 0
        except Exception:
            if os.environ.get("XRL_DIST_DEBUG"):
                import traceback
                traceback.print_exc()
            ok = False
        mine = torch.cat([grad, params])
        ref = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(ref, mine)                               # every rank reports the same averaged gradient and parameters
        ok = ok and all(torch.equal(r, mine) for r in ref)
        local = [slabs_all[k, r].to(device).double().sum(0).float() for r in range(world)]    # the launch's float64 slab sums
        expect = local[0]
        for r in range(1, world):
            expect = expect + local[r]
        ok = ok and bool(torch.equal(grad, expect * (1.0 / world)))
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok))
    try:
        xc.close()
    except Exception:
        pass
    return all(oks)


def exchange_usable(device):
    """May gradients be averaged inside the optimiser launch (GradientExchange)?  More than one rank, every rank's GPU
    reachable from this one (IPC + peer access), not disabled by XRL_DIST_EXCHANGE=0, proven once by exchange_selftest."""
    global _exchange_ok
    if _exchange_ok is None:
        _exchange_ok = False
        if dist.is_initialized() and 1 < dist.get_world_size() <= 8 and torch.cuda.is_available() and \
                os.environ.get("XRL_DIST_EXCHANGE", "1") != "0":
            _exchange_ok = exchange_selftest(device)
    return _exchange_ok


def broadcast_(flat, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def shard_range(n_total, rank=None, world=None):
    """Contiguous shard [lo, hi) of n_total independent envs for this rank."""
    rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
    world = world_size() if world is None else world
    per = n_total // world
    return rank * per, (rank + 1) * per


def barrier():
    if dist.is_initialized():
        dist.barrier()

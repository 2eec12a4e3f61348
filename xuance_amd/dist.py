"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The path shards over environments: every rank owns n_envs/world envs and its own rollout
buffer, parameters are replicated, and the ONLY data-path collective is one all-reduce (mean) of the flat
gradient buffer per optimiser step -- 136 KB..570 KB, latency-bound on the xGMI mesh, hence a single flat message
instead of DDP's per-bucket calls -- plus, when observation normalisation is on, the moments all-reduce of
xuance/torch/utils/tensor_statistics.py:48-58.
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
    if not dist.is_initialized() or dist.get_world_size() == 1:
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

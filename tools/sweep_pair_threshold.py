"""Scratch: where 64-row tiles start to pay now that they run the split-product kernel -- update phase of the headline loop for
n_envs in {32 ... 256} (minibatches of 1 024 ... 8 192 rows = 32 ... 256 32-row tiles) with 32-row tiles (float32 instruction) and
with 64-row tiles (split products).  Prints ms per update phase (64 minibatches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

for n in (32, 48, 64, 96, 128, 160, 192, 256):
    row = []
    for pair in (False, True):
        cfg = bench.make_config(n, 256, 1, 0)
        cfg.use_pair_update = pair
        torch.manual_seed(1)
        agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
        for _ in range(3):
            agent.rollout(); agent.update()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(7):
            agent.rollout(); torch.cuda.synchronize()
            e0.record(); agent.update(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        row.append(ts[len(ts) // 2])
        assert agent.learner.pair == pair and (agent.learner.frag16 is not None) == pair
    print("n_envs %3d  tiles/minibatch %3d   32-row tiles %.3f ms   64-row tiles (split products) %.3f ms" % (n, n * 256 // 8 // 32, row[0], row[1]), flush=True)

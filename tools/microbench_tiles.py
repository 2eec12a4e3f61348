"""Scratch: update-phase time of the CartPole class per minibatch-kernel variant and minibatch size (the any-shape ppo_fused_kernel: one workgroup
per 32-row tile; ppo_trunk_kernel with 32- / 64-row tiles: (tile, role) workgroups)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
for n in (16, 32, 64, 128, 256):
    row = []
    for name, split, pair in (("any-shape", False, False), ("trunk32", True, False), ("trunk64", True, True)):
        cfg = bench.make_config(n, 256, 1, 0); cfg.use_role_split_update = split; cfg.use_pair_update = pair
        torch.manual_seed(1)
        agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
        for _ in range(2):
            agent.rollout(); agent.update()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            agent.update()
        torch.cuda.synchronize()
        row.append("%s %.3f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
    print("n_envs %3d (%3d tiles per minibatch): update phase  " % (n, n), "  ".join(row), flush=True)

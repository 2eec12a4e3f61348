"""Scratch: does the whole-rollout launch's time depend on the allocation (new agent, same process) or on the process?  Five
agents in a row, each timed over 30 rollouts after 3 warm-ups."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
n = 256
keep = []
for k in range(5):
    cfg = bench.make_config(n, 256, 1, 0)
    torch.manual_seed(1)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
    for _ in range(3):
        agent.rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        agent.rollout()
    torch.cuda.synchronize()
    print("agent", k, "rollout ms", round((time.perf_counter() - t0) / 30 * 1e3, 4), "status", agent.persist_status.tolist() if getattr(agent, "persist_status", None) is not None else None)
    keep.append(agent)          # (keep the allocations: the next agent gets other addresses)

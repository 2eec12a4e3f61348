"""Scratch diagnostics: 1 500 rollout + update iterations of the headline loop and 300 of the C4 loop; the whole-rollout launches' status
words every few hundred iterations (a time-out or a placement fallback would show as a non-zero word)."""
import sys, time; sys.path.insert(0, ".")
import torch, bench
from tools import bench_workloads as bw
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv, SyntheticMujocoVecEnv
cfg = bench.make_config(256, 256, 1, 0)
agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(256, seed=1))
t0 = time.time()
for i in range(1500):
    agent.rollout(); agent.update()
    if i % 250 == 0:
        torch.cuda.synchronize(); print("c2", i, agent.persist_status.tolist(), round(time.time() - t0, 1), flush=True)
torch.cuda.synchronize(); print("c2 done", agent.persist_status.tolist(), agent.learner.read_optimizer().step)
from argparse import Namespace
c4 = bw.c4_config(128, 256, 1, 0) if hasattr(bw, "c4_config") else None
if c4 is not None:
    a4 = PPO_Agent(c4, SyntheticMujocoVecEnv(128, seed=4))
    for i in range(300):
        a4.rollout(); a4.update()
        if i % 100 == 0:
            torch.cuda.synchronize(); print("c4", i, a4._wr_status.tolist(), flush=True)
    torch.cuda.synchronize(); print("c4 done", a4._wr_status.tolist())

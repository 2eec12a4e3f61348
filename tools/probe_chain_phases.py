"""Scratch: where the prologue of a chained minibatch launch (xrl_ppo_trunk_chained, csrc/opt_chain.h) spends its time: 100 MHz
real-time stamps of every workgroup (start | slabs summed | B1 flag out | past B1 | Adam stored | released | past B2 poll |
acquired) and the workgroup's start / end stamps of the minibatch body."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
n = 256
cfg = bench.make_config(n, 256, 1, 0); cfg.use_chained_update = True; cfg.use_hip_graph = False
torch.manual_seed(1)
agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
agent.rollout(); agent.update(); torch.cuda.synchronize()
lr, mem = agent.learner, agent.memory
dbg = torch.zeros(2048 + 8 * 512, dtype=torch.int64, device="cuda")
lr._dbg = dbg
res = []
for rep in range(3):
    for k in range(6):
        lr.enqueue_minibatch_fused(mem, agent.idx[k], lr.stats[k], defer=True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    c = d[2048:2048 + 8 * 256].reshape(256, 8).astype(np.float64) * 1e-2        # us
    se = d[16:16 + 2 * 256].reshape(256, 2).astype(np.float64) * 1e-2
    t0 = c[:, 0].min()
    nw = 134
    names = ["start", "slabs summed", "B1 flag out", "past B1", "Adam stored", "released", "past B2 poll", "acquired"]
    print("rep", rep)
    for j, nm in enumerate(names):
        col = c[:nw, j] if j in (3, 4, 5) else c[:, j]
        print("  %-14s workers: min %6.2f med %6.2f max %6.2f   | all: min %6.2f med %6.2f max %6.2f" %
              (nm, (c[:nw, j] - t0).min(), np.median(c[:nw, j] - t0), (c[:nw, j] - t0).max(), (col - t0).min(), np.median(col - t0), (col - t0).max()))
    print("  body end: min %.2f med %.2f max %.2f us after the first start" % ((se[:, 1] - t0).min(), np.median(se[:, 1] - t0), (se[:, 1] - t0).max()))
    res.append({"names": names, "workers_median_us": [float(np.median(c[:nw, j] - t0)) for j in range(8)],
                "all_max_us": [float((c[:, j] - t0).max()) for j in range(8)], "body_end_max_us": float((se[:, 1] - t0).max())})
    lr.finish_pending(); torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06_b_chain_phases.json"), "w"), indent=1)

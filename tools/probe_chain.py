"""Scratch: update phase of the headline (256 envs x 256, 64 minibatches of 8 192) with the optimiser step chained into the next
minibatch launch (xrl_ppo_trunk_chained) against the launch pair {minibatch, xrl_reduce_adam}; wall time of the captured phase."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
out = {}
for n in (256, 128):
    for chained in (False, True, False, True):
        cfg = bench.make_config(n, 256, 1, 0); cfg.use_chained_update = chained
        if n == 128:
            cfg.use_pair_update = True          # 64-row tiles: 128 workgroups < 134 -> not chained; 32-row tiles: 256 workgroups
            cfg.use_pair_update = False
        torch.manual_seed(1)
        agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
        for _ in range(2):
            agent.rollout(); agent.update()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            agent.update()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        used = any(agent.learner._chain_ok.values())
        print("n_envs %d chained=%s (used: %s): update phase %.3f ms = %.2f us per minibatch; sync[2]=%d"
              % (n, chained, used, ms, ms * 1e3 / 64, int(agent.learner.opt_sync[2].item())), flush=True)
        out.setdefault(str(n), []).append({"chained": chained, "used": used, "update_ms": ms})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_a_probe_chain.json"), "w"), indent=1)

"""Where a vector step of the actor rollout kernel (csrc/rollout_actor.hip) spends its time: shader-clock stamps of workgroup 0's
chain wave at step T / 2 (xrl_rollout_run_t.dbg), and the HIP-event time of the rollout launches (actor kernel + batched values
launch) for 256 and 16 envs, whole-rollout and per-step forms.  Writes gpurun_out/rollout_phases.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

names = {1: "partial sums of all workgroups collected", 2: "statistics merged"}
out = {}
for n in (256, 16):
    cfg = bench.make_config(n, 256, 1, 0)
    torch.manual_seed(1)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
    agent.rollout(); agent.update(); agent.rollout(); torch.cuda.synchronize()
    assert agent._actor_rollout() is not None and agent.persist_status is not None
    dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
    agent.rollout_dbg = dbg
    agent._enqueue_rollout_fused(kernel_only=True); torch.cuda.synchronize()
    d = dbg.tolist()
    agent.rollout_dbg = None
    us_pair = bench._event_time_us(lambda: agent._enqueue_rollout_fused(kernel_only=True), 10)
    cpr = agent._actor_rollout()
    us_actor = bench._event_time_us(lambda: cpr.run(0, 256), 10)
    us_values = bench._event_time_us(lambda: cpr.values(0, 256), 10)
    us_step1 = bench._event_time_us(lambda: cpr.run(5, 1), 50)
    ph = {"partial sums of all workgroups collected (from the loop top)": d[1] - d[0], "statistics merged, up to barrier 1": d[2] - d[1],
          "barrier 1 -> logits ready (barrier 3)": d[4] - d[2], "action sampled, fate + record of the env in LDS": d[5] - d[4],
          "barrier 0 + new partial sums reduced + message stored": d[6] - d[5],
          # matrix wave 0 (d[8] just behind barrier 1)
          "matrix wave 0: normalise + first layer (8 MFMA) + 64 MFMA issued": d[9] - d[8],
          "matrix wave 0: epilogue (activation, 8 head MFMAs, LDS)": d[10] - d[9],
          "physics wave: next step's physics for both actions": d[12] - d[11],
          "physics wave done, after the chain wave reached barrier 0": d[12] - d[5]}
    out[str(n)] = {"actor+values_us": round(us_pair, 1), "actor_kernel_us": round(us_actor, 1), "values_kernel_us": round(us_values, 1),
                   "one_step_launch_us": round(us_step1, 2), "us_per_step": round(us_actor / 256, 3),
                   "phase_cycles_of_step_128": ph, "stamped_cycles": d[6] - d[0], "status": agent.persist_status.tolist()}
    print(n, json.dumps(out[str(n)]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rollout_phases.json"), "w"), indent=1)

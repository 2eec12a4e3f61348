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

names = ["loop top", "partial sums of all workgroups collected", "statistics merged", "first layer done (barrier 2)",
         "logits / physics / draws ready (barrier 3)", "action sampled", "new partial sums published", "records stored"]
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
    ph = {names[i + 1]: d[i + 1] - d[i] for i in range(7)}
    out[str(n)] = {"actor+values_us": round(us_pair, 1), "actor_kernel_us": round(us_actor, 1), "values_kernel_us": round(us_values, 1),
                   "one_step_launch_us": round(us_step1, 2), "us_per_step": round(us_actor / 256, 3),
                   "phase_cycles_of_step_128": ph, "stamped_cycles": d[7] - d[0], "status": agent.persist_status.tolist()}
    print(n, json.dumps(out[str(n)]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rollout_phases.json"), "w"), indent=1)

"""Scratch: the PPO-Atari line with the acting tail as one launch (xrl_ppo_act_tail) against the launches (config.use_frame_act_tail)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_secondary as bs
import xuance_amd.agents.ppo_agent as pa
out = []
orig = pa._get
for tail in (False, True, False, True):
    pa._get = lambda cfg, name, default=None, _t=tail: (_t if name == "use_frame_act_tail" else orig(cfg, name, default))
    r = bs.ppo_atari()
    print("tail", tail, r["value"], {k: r[k] for k in r if "ms" in k}, flush=True)
    out.append({"tail": tail, "value": r["value"], "ms_per_step": r["ms_per_step"], "rollout_ms": r.get("rollout_ms"), "update_ms": r.get("update_ms")})
pa._get = orig
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_h_ppo_atari_tail.json"), "w"), indent=1)

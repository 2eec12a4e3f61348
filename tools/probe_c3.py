"""Scratch: DQN-C3 and PPO-Atari lines with / without the convolution weight gradients on a side branch (ConvStack.overlap_wgrad)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_secondary as bs
from xuance_amd import nets
from xuance_amd._lib import call
out = {}
for mixed in (0, 1, 0, 1):
    call("xrl_set_conv_dw_mixed", mixed)
    r = bs.dqn_c3(steps=200, buffer_size=32768, start_training=2000)
    print("one dW launch for all kinds", mixed, "dqn_c3", r["value"], r.get("update_us"), flush=True)
    out.setdefault("dqn_c3_mixed_dw", []).append({"mixed": mixed, "value": r["value"], "update_us": r.get("update_us")})
for mixed in (0, 1):
    call("xrl_set_conv_dw_mixed", mixed)
    r = bs.ppo_atari()
    print("one dW launch for all kinds", mixed, "ppo_atari", r["value"], {k: r[k] for k in r if "ms" in k}, flush=True)
    out.setdefault("ppo_atari_mixed_dw", []).append({"mixed": mixed, "value": r["value"], "ms_per_step": r.get("ms_per_step"), "update_ms": r.get("update_ms")})
call("xrl_set_conv_dw_mixed", 0)
for ov in (False, True):
    nets.ConvStack.overlap_wgrad = ov
    r = bs.dqn_c3(steps=200, buffer_size=32768, start_training=2000)
    print("overlap", ov, "dqn_c3", r["value"], r.get("update_us"), flush=True)
    out.setdefault("dqn_c3", []).append({"overlap": ov, "value": r["value"], "update_us": r.get("update_us")})
for ov in ():
    nets.ConvStack.overlap_wgrad = ov
    r = bs.ppo_atari()
    print("overlap", ov, "ppo_atari", r["value"], r.get("ms_per_step"), {k: r[k] for k in r if "ms" in k}, flush=True)
    out.setdefault("ppo_atari", []).append({"overlap": ov, "value": r["value"], "ms_per_step": r.get("ms_per_step")})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_g_conv_overlap.json"), "w"), indent=1)

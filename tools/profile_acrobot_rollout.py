"""Kernel table of the Acrobot-v1 PPO loop (bench_secondary.ppo_acrobot) for rocprofv3 --kernel-trace --stats."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_secondary as bs, bench
tail = not (len(sys.argv) > 1 and sys.argv[1] == "layered")
def mk(*a):
    c = bench.make_config(*a)
    c.use_device_act_tail = tail
    return c
r = bs.ppo_acrobot(mk, steps=3, warmup=1)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "rollout_ms", "update_ms")}))

"""Whole agent loops of the off-policy configs (act -> env provider -> store -> update phase) for
rocprofv3 --kernel-trace --stats: `python tools/profile_offpolicy_loop.py dqn_c3|qmix_c5|qmix_rnn [steps]`.
Prints wall time per vector step; the kernel table gives the device-busy time to hold against it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_offpolicy as B

which, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.perf_counter()
out = {"dqn_c3": B.dqn_c3, "qmix_c5": B.qmix_c5, "qmix_rnn": lambda s: B.qmix_c5_rnn(s, False)}[which](steps)
print({k: v for k, v in out.items() if k != "config"}, "total_s", round(time.perf_counter() - t0, 2))

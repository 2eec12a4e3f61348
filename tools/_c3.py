import sys, json; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench_secondary as bs
for i in range(3):
    r = bs.dqn_c3()
    print(json.dumps({k: r[k] for k in ("value", "vector_step_us", "update_us")}), flush=True)

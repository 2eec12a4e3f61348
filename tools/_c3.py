import sys, json; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench_secondary as bs
from xuance_amd.agents import DQN_Agent
import bench
_init = DQN_Agent.__init__
mode = {"fused": True}
def init(self, *a, **k):
    _init(self, *a, **k)
    if not mode["fused"]:
        self._act_fused = False
DQN_Agent.__init__ = init
print(json.dumps(bench.box_yardstick()))
for i in range(4):
    mode["fused"] = i % 2 == 0
    r = bs.dqn_c3()
    print(mode, json.dumps({k: r[k] for k in ("value", "vector_step_us", "update_us")}), flush=True)

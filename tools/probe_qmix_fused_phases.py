"""DIAGNOSTICS -- cycle-counter stamps of workgroup 0 of xrl_qmix_fused_update at its phase boundaries (xrl_qmix_fused_t.dbg):
1 = weights + inputs staged, 4 = forward done, 6 = mixing / TD done, 8 = backward done.
`python tools/probe_qmix_fused_phases.py [items_per_wg]` on a GPU box."""
import sys
sys.path.insert(0, "/root/repo")
from tools import bench_secondary as bs
import torch
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv
c = bs._qmix_cfg(64, False); c.fused_qmix_items_per_wg = int(sys.argv[1]) if len(sys.argv) > 1 else None; c.use_hip_graph = False
c.fused_qmix_products = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agent = QMIX_Agents(c, SyntheticSMACVecEnv(64, seed=3))
agent.train(30)
lr = agent.learner
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
lr._fused.struct.dbg = dbg.data_ptr()
for _ in range(2):
    lr._step(32)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    print("items", lr._fused.items_per_wg, "products", lr._fused.struct.products, {k: int(d[k] - d[0]) for k in (1, 4, 6, 8)})
    print("   forward phases end (all waves through the barrier):", [int(d[16 + l] - d[1]) for l in range(5)])
    for l in range(5):
        print("   phase", l, "wave arrival at the barrier, from the phase's start:", [int(d[32 + 16 * l + w] - (d[1] if l == 0 else d[16 + l - 1])) for w in range(16)])
import bench
lr._fused.struct.dbg = None
print("fused update launch(es) + optimiser: %.2f us per update" % bench._event_time_us(lambda: lr._step(32), 50))

"""Scratch: first difference between the phase launch and the launch pairs (tests/test_gpu_offpolicy_agents.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_secondary as bs
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv
ag = []
for phase in (True, False):
    cfg = bs._qmix_cfg(64, False); cfg.use_qmix_phase_launch = phase; cfg.sync_frequency = 50
    torch.manual_seed(0)
    ag.append(QMIX_Agents(cfg, SyntheticSMACVecEnv(64, seed=3)))
a, b = ag
for k in range(14):
    a.train(1); b.train(1); torch.cuda.synchronize()
    la, lb = a.learner, b.learner
    d = lambda x, y: float((x.double() - y.double()).abs().max())
    st = (la.read_optimizer().step, lb.read_optimizer().step)
    print(k, "steps", st, "params", d(a.model.params.flat, b.model.params.flat), "grad", d(la.optimizer.grad, lb.optimizer.grad),
          "m", d(la.optimizer.m, lb.optimizer.m), "v", d(la.optimizer.v, lb.optimizer.v), "target", d(a.model.target_flat, b.model.target_flat),
          "img", d(la._fused.img_eval, lb._fused.img_eval) if getattr(la, "_fused", None) is not None and getattr(lb, "_fused", None) is not None else None,
          "sums", d(la._epoch_sums, lb._epoch_sums) if hasattr(la, "_epoch_sums") and hasattr(lb, "_epoch_sums") else None, flush=True)
    if st[0] > 0 and d(la.optimizer.grad, lb.optimizer.grad) > 0:
        g = (la.optimizer.grad - lb.optimizer.grad).abs()
        nz = torch.nonzero(g).flatten()
        print("   loss sums per update (phase):", la._epoch_sums[:, 0].tolist())
        print("   loss sums per update (pairs):", lb._epoch_sums[:, 0].tolist())
        print("   drawn rows equal:", torch.equal(la._idx, lb._idx), "scalars", la._phase_scalars.tolist())
        st_b = lb.read_optimizer()
        print("   differing gradient elements:", nz.numel(), "first", nz[:8].tolist(), "last", nz[-8:].tolist(), "P", a.model.params.P)
        break

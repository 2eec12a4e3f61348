"""Scratch: the feed-forward QMIX update phase (8 updates per vector step) as one launch (xrl_qmix_fused_phase) against the captured
sequence of 8 x {xrl_qmix_fused_update, xrl_reduce_adam}: time per update and the loop's env-steps/s."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_secondary as bs
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv
out = []
for phase in (False, True, False, True):
    cfg = bs._qmix_cfg(64, False); cfg.use_qmix_phase_launch = phase
    torch.manual_seed(0)
    agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(64, seed=3))
    agent.train(20)
    bs._settle()
    s0, t0 = agent.current_step, time.perf_counter()
    while True:
        agent.train(200); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= 0.5:
            break
    lr = agent.learner
    if lr._phase_launch:
        dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
        lr._fused.struct.dbg = dbg.data_ptr()
        lr._buf_enqueue(); torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        print("  update %d of the phase, workgroup 0 (cycles from its start): staged %d, forward %d, mixing %d, backward %d | meeting A +%d, slab sums +%d, loss sums + barrier +%d, Adam +%d, meeting B +%d"
              % (4, d[1] - d[0], d[4] - d[0], d[6] - d[0], d[8] - d[0], d[65] - d[64], d[66] - d[65], d[67] - d[66], d[68] - d[67], d[69] - d[68]))
        lr._fused.struct.dbg = None
    us = bs._events_us(lr._buf_graph.launch, 200) / 8
    r = {"phase_launch": bool(lr._phase_launch), "update_us": round(us, 2), "env_steps_per_s": round((agent.current_step - s0) / dt, 1),
         "sync_fail": int(lr._phase_sync[2].item()), "xcc_mask": int(lr._phase_sync[3].item())}
    print(r, flush=True); out.append(r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_d_qmix_phase.json"), "w"), indent=1)

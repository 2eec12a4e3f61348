"""Scratch diagnostics: repeat the same seeded rollout -> update -> rollout and report the first quantity that differs
between repetitions (a race shows up as a run-to-run difference)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_headline import c2_config, npy
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

n, T, reps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 256, 5)
kw = dict(use_persistent_rollout=os.environ.get("PERSIST", "1") == "1", use_hip_graph=os.environ.get("GRAPH", "1") == "1",
          use_fused_optimizer=os.environ.get("FUSEDOPT", "1") == "1",
          persistent_coherent_exchange=os.environ.get("COHERENT", "0") == "1")
if len(sys.argv) > 4:
    kw.update(n_epochs=int(sys.argv[4]), n_minibatch=int(sys.argv[5]))
ref, bad = None, 0
for r in range(reps):
    torch.manual_seed(0)
    env = DeviceCartPoleVecEnv(n, seed=3)
    agent = PPO_Agent(c2_config(n, T, **kw), env)
    snaps = []
    for it in range(3):
        agent.rollout(); torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        snaps.append(("rollout%d" % it, f))
        agent.update(); torch.cuda.synchronize()
        snaps.append(("update%d" % it, {"params": npy(agent.model.params.flat), "m": npy(agent.learner.optimizer.m)}))
    if ref is None:
        ref = snaps
        continue
    for (name, a), (_, b) in zip(ref, snaps):
        diff = [k for k in a if not np.array_equal(a[k], b[k])]
        if diff:
            bad += 1
            k = diff[0]
            w = np.argwhere(a[k] != b[k])
            print(f"rep {r}: first difference in {name}: fields {diff}; {k} differs at {len(w)} places, first {w[:4].tolist()} "
                  f"ref {a[k][tuple(w[0])]} now {b[k][tuple(w[0])]}", flush=True)
            break
print(f"n={n} T={T} {kw}: {bad} of {reps - 1} repetitions differed", flush=True)

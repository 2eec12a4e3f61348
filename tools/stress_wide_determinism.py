"""Scratch diagnostics: repeat the same seeded C4-shape rollout -> update chain (whole-rollout launch, split weight gradient) and report
run-to-run differences and the rollout's status words."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from argparse import Namespace
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import SyntheticMujocoVecEnv
n, T, reps = 128, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 6
ref, bad = None, 0
for r in range(reps):
    cfg = Namespace(agent="PPO", representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                    critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", seed=1, parallels=n,
                    running_steps=10 ** 9, horizon_size=T, n_epochs=2, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25,
                    ent_coef=0.0, clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                    grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                    distributed_training=False, device="cuda", model_dir="/tmp/x", use_hip_graph=True)
    torch.manual_seed(0)
    agent = PPO_Agent(cfg, SyntheticMujocoVecEnv(n, seed=4))
    snaps = []
    for it in range(4):
        agent.rollout(); torch.cuda.synchronize()
        snaps.append({k: v.cpu().numpy().copy() for k, v in agent.memory.soa.fields.items()})
        agent.update(); torch.cuda.synchronize()
        snaps.append({"params": agent.model.params.flat.cpu().numpy().copy()})
    st = agent._wr_status.tolist() if getattr(agent, "_wr_status", None) is not None else None
    if ref is None:
        ref = snaps
    else:
        for i, (a, b) in enumerate(zip(ref, snaps)):
            for k in a:
                if not np.array_equal(a[k], b[k]):
                    bad += 1
                    print("rep", r, "snapshot", i, k, "differs: max", float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()))
                    break
    print("rep", r, "status", st)
print("differences:", bad)

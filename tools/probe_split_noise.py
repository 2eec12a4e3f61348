"""Scratch: per-tensor distance of ONE minibatch gradient of the C2 fixture (8 192 rows, the reference's own initial parameters and batch)
from the reference's float64 twin, for the minibatch kernels side by side: float32 matrix instruction (csrc/ppo_trunk.hip), 3-way bf16
split products with the wave pairs splitting rows / k (csrc/ppo_trunk_bx.hip).  Unclipped, unreduced: slabs summed in float64 here."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from argparse import Namespace
import numpy as np
import torch
from conftest import load_golden, sub
from test_gpu_ppo import _load_rows
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

size, n, T = "c2", 32, 256
g = load_golden(f"ppo_categorical_{size}")
lr, vf, ent, clip, gclip, ef, total = g["cfg"]
res = {}
for tag, split, ks in (("f32", False, 1), ("bx-rows", True, 0), ("bx-kb", True, 1), ("bx-k", True, 2)):
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128], critic_hidden_size=[128],
                    activation="leaky_relu", seed=1, parallels=n, running_steps=int(total) * n * T, horizon_size=T, n_epochs=1,
                    n_minibatch=1, learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip),
                    gamma=0.98, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=float(gclip),
                    end_factor_lr_decay=float(ef), use_obsnorm=False, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_models", use_hip_graph=False,
                    use_role_split_update=True, use_pair_update=True, use_split_products=split)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
    mem, lr_ = agent.memory, agent.learner
    agent.model.load_state_dict(sub(g, "init"))
    idx = torch.arange(n * T, dtype=torch.int64, device="cuda").view(1, -1)
    lr_.prepare_fused(mem, n * T)
    lr_.prepare_rows(idx.numel())
    _load_rows(mem, sub(g, "u0/batch"), n, T)
    lr_.refresh_fused_params(mem, idx)
    ops.set_split_product_ksplit(ks)
    lr_.enqueue_minibatch_fused(mem, idx[0], None, finish=False)
    torch.cuda.synchronize()
    ops.set_split_product_ksplit(1)
    P = agent.model.params.P
    sl = lr_.fslabs[:n * T // 64].double().sum(0)
    gsum = sl[:P].clone()
    fold = lr_.fold
    gsum[:fold[1]] += sl[fold[0]:fold[0] + fold[1]]
    res[tag] = gsum.cpu().numpy()
    offs, shapes = agent.model.params.offsets, agent.model.params.shapes
print("%-34s %12s %12s %12s %12s %12s   (max |g - g64_unclipped| / max|g64|; the reference's float32: last column)" % ("tensor", *res, "ref f32"))
# the fixture's gradients are CLIPPED (scaled by c = max_norm / norm): compare directions after scaling ours by the same factor
names = list(sub(g, "u0/grad"))
g64 = {k: g[f"u0/grad64/{k}"].astype(np.float64) for k in names}
g32 = {k: g[f"u0/grad/{k}"].astype(np.float64) for k in names}
for k in names:
    lo = offs[k]; hi = lo + int(np.prod(shapes[k]))
    r64 = g64[k].reshape(-1)
    S = np.abs(r64).max()
    row = []
    for tag in res:
        mine = res[tag][lo:hi]
        c = float(np.dot(mine, r64) / np.dot(mine, mine))        # the clip factor (least squares; the same for every tensor up to noise)
        row.append(np.abs(mine * c - r64).max() / S)
    row.append(np.abs(g32[k].reshape(-1) - r64).max() / S)
    print("%-34s %12.3e %12.3e %12.3e %12.3e %12.3e" % (k, *row))
d = np.abs(res["bx-kb"] - res["f32"]); dr = np.abs(res["bx-rows"] - res["f32"])
for k in names:
    lo = offs[k]; hi = lo + int(np.prod(shapes[k]))
    S = np.abs(res["f32"][lo:hi]).max()
    print("vs f32 kernel %-30s bx-rows %.3e  bx-kb %.3e  (of max|g| %.3e)" % (k, dr[lo:hi].max() / S, d[lo:hi].max() / S, S))

"""TEST DIAGNOSTICS (uses the oracle; run by hand on a GPU box: python tests/diagnostics/grad_noise.py) -- rounding noise of ONE minibatch gradient (8 192 rows, fixture ppo_categorical_c2) -- the HIP learner
(layered path and the fused minibatch kernel) and the float32 oracle, each against the float64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from argparse import Namespace
from conftest import load_golden, sub
from oracle import xrl_oracle as o
from xuance_amd.nets import ActorCriticNet
from xuance_amd.learners import PPO_Learner

g = load_golden("ppo_categorical_c2")
lr, vf, ent, clip, gclip, ef, total = g["cfg"]
cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.98, learning_rate=float(lr),
                vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip), use_grad_clip=False, grad_clip_norm=float(gclip),
                end_factor_lr_decay=float(ef), distributed_training=False, device="cuda", model_dir="/tmp/x")
net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu")
net.load_state_dict(sub(g, "init"))
learner = PPO_Learner(cfg, net, None)
b = sub(g, "u0/batch")
learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"], advantages=b["advantages"],
               aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
torch.cuda.synchronize()
ours = {k: net.params.view(k, learner.optimizer.grad).cpu().numpy().astype(np.float64) for k in net.ref_order}
c = dict(vf_coef=vf, ent_coef=ent, clip_range=clip)
sd32 = {k: v.copy() for k, v in sub(g, "init").items()}
_, g32 = o.ppo_forward_backward(sd32, b, c)
sd64 = {k: v.astype(np.float64) for k, v in sub(g, "init").items()}
_, g64 = o.ppo_forward_backward(sd64, {k: np.asarray(v, np.float64) for k, v in b.items()}, c)
ref = sub(g, "u0/grad")       # torch (reference) gradient, clipped: compare direction only where unclipped... skip scale: use ratio
print("%-28s %10s %10s %10s %10s %10s" % ("tensor", "|g|rms", "hip-f64", "np32-f64", "torch-f64*", "hip/np32"))
tn = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g64.values()))
coef = min(gclip / (tn + 1e-6), 1.0)
for k in net.ref_order:
    e_h = np.sqrt(np.mean((ours[k] - g64[k]) ** 2)); e_n = np.sqrt(np.mean((g32[k].astype(np.float64) - g64[k]) ** 2))
    e_t = np.sqrt(np.mean((ref[k].astype(np.float64) - coef * g64[k]) ** 2)) / coef
    print("%-28s %10.3e %10.3e %10.3e %10.3e %10.2f" % (k, np.sqrt(np.mean(g64[k] ** 2)), e_h, e_n, e_t, e_h / max(e_n, 1e-30)))

"""Scratch: time the minibatch kernel alone (graph of 64 launches) -- ppo_fast vs ppo_split (both roles / one role)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

split = os.environ.get("SPLIT", "1") == "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = bench.make_config(n, 256, 1, 0)
cfg.use_role_split_update = split
torch.manual_seed(1)
agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
for _ in range(2):
    agent.rollout(); agent.update()
torch.cuda.synchronize()
lr, mem, m = agent.learner, agent.memory, agent.model
f, bs, T, nb = mem.soa.fields, agent.batch_size, 256, agent.idx.shape[0]

def mb(k):
    ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                            f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                            f_logp=f["aux_old_logp"], idx=agent.idx[k], stats=lr.stats[k], slabs=lr.fslabs,
                            partials=lr.fpartials, diag=None, slab_stride=lr.slab_stride, l0_fold_off=lr.fold[0] if lr.fold else 0,
                            M=bs, n_envs=n, T=T, D=4, frag_image=lr.frag, f_packed=lr.packed,
                            f_rows=lr.rows[k * bs * 8:(k + 1) * bs * 8], A=m.action_dim, clip_range=lr.clip_range,
                            vf_coef=lr.vf_coef, ent_coef=lr.ent_coef)
opt = lr.optimizer
clip = lr.grad_clip_norm
def ra():
    ops.reduce_adam(lr.fslabs, lr.n_tiles, lr.slab_stride, m.params.flat, opt.grad, opt.m, opt.v, m.params.P, opt.state, lr.sumsq,
                    clip, lr._mirrors, lr.opt_sync, fold=lr.fold)
g_mb, g_ra, g_both = ops.Graph(), ops.Graph(), ops.Graph()
torch.cuda.synchronize()
with g_mb:
    for k in range(nb): mb(k)
with g_ra:
    for k in range(nb): ra()
with g_both:
    for k in range(nb): mb(k); ra()
for g in (g_mb, g_ra, g_both): g.launch()
out = {"split": split, "only_role": os.environ.get("XRL_SPLIT_ONLY_ROLE"), "n_envs": n,
       "mb_alone_us": round(bench._event_time_us(g_mb.launch, 5) / nb, 2), "reduce_adam_alone_us": round(bench._event_time_us(g_ra.launch, 5) / nb, 2),
       "both_us": round(bench._event_time_us(g_both.launch, 5) / nb, 2)}
print(json.dumps(out))

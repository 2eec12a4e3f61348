"""Scratch: phase stamps (shader cycles from kernel start, last workgroup) of ppo_fast_kernel / ppo_split_kernel.
Needs a probe build:  XRL_BUILD_DEFINES=-DXRL_TILE_PROBE python -m xuance_amd.build --force"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for split in (False, True):
    cfg = bench.make_config(n, 256, 1, 0); cfg.use_role_split_update = split
    torch.manual_seed(1)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
    agent.rollout(); agent.update(); torch.cuda.synchronize()
    lr, mem, m = agent.learner, agent.memory, agent.model
    f, bs = mem.soa.fields, agent.batch_size
    for who in ((0,) if not split else (0, 77)):
        dbg = torch.zeros(16, dtype=torch.int64, device="cuda"); dbg[15] = who
        for _ in range(3):
            ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                                    f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                                    f_logp=f["aux_old_logp"], idx=agent.idx[3], stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                                    diag=None, slab_stride=lr.slab_stride, l0_fold_off=lr.fold[0] if lr.fold else 0, M=bs, n_envs=n, T=256,
                                    D=4, A=2, clip_range=0.2, vf_coef=0.25, ent_coef=0.01, dbg=dbg, frag_image=lr.frag, f_packed=lr.packed,
                                    f_rows=lr.rows[3 * bs * 8:4 * bs * 8])
            torch.cuda.synchronize()
        d = dbg.tolist()
        if split:
            names = ["start", "rows gathered", "h1", "h2 (fwd MFMA)", "heads+loss+g2", "small grads", "dW1", "dH1+g1", "end"]
            print("split", "actor" if who == 77 else "critic", dict(zip(names, d[:9])))
        else:
            names = ["start", "h1 ready", "h2 ready", "heads+loss done", "small grads done", "dW done", "dH partials", "g1 ready", "end", "rows gathered"]
            print("fast ", dict(zip(names, d[:10])))

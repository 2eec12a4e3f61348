"""Diagnostics: where the in-loop time of the headline's [minibatch launch, optimiser launch] pair goes.  HIP-event times of
  (a) the minibatch launch alone, repeated; (b) the optimiser launch alone, repeated; (c) the pair, alternating (the real sequence);
  (d) the pair with the minibatch launch's dW1 stores switched off (xrl_ppo_fused_t.pad3 bit 0: wrong numbers, same everything else) --
      if (c) - (d) is much more than the stores' own time, the optimiser launch is waiting for the write-back of the gradient slabs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
n = 256
cfg = bench.make_config(n, 256, 1, 0)
torch.manual_seed(1)
agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
agent.rollout(); agent.update(); torch.cuda.synchronize()
lr, mem, m = agent.learner, agent.memory, agent.model
f, bs, opt = mem.soa.fields, agent.batch_size, agent.learner.optimizer

def trunk(pad3=0):
    ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                            f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                            f_logp=f["aux_old_logp"], idx=agent.idx[3], stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                            diag=None, slab_stride=lr.slab_stride, l0_fold_off=lr.fold[0] if lr.fold else 0, M=bs, n_envs=n, T=256,
                            D=4, A=2, clip_range=0.2, vf_coef=0.25, ent_coef=0.01, frag_image=lr.frag, f_packed=lr.packed,
                            f_rows=lr.rows[3 * bs * 8:4 * bs * 8], pad0=64 if lr.pair else 0, pad3=pad3)

def adam():
    ops.reduce_adam(lr.fslabs, lr.n_slabs, lr.slab_stride, m.params.flat, opt.grad, opt.m, opt.v, m.params.P, opt.state, lr.sumsq, 0.5,
                    lr._mirrors, lr.opt_sync, fold=lr.fold)

def pair(pad3=0):
    trunk(pad3); adam()

out = {}
for name, fn, reps in (("minibatch launch alone", trunk, 64), ("optimiser launch alone", adam, 64), ("pair, alternating", pair, 64),
                       ("pair, minibatch launch without its dW1 stores", lambda: pair(1), 64), ("minibatch launch alone, without its dW1 stores", lambda: trunk(1), 64)):
    for _ in range(2):
        t = bench._event_time_us(fn, reps)
    out[name + " (us)"] = round(t, 2)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pair_sequence.json"), "w"), indent=1)

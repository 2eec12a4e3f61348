"""Phase timing (shader-clock stamps of the last workgroup) of the fused PPO minibatch kernel at the bench config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
from bench import make_config

n = 256
agent = PPO_Agent(make_config(n, 256, 1, 0), DeviceCartPoleVecEnv(n, seed=1))
agent.rollout(); agent.update(); torch.cuda.synchronize()
lr, mem = agent.learner, agent.memory
m, f = lr.model, mem.soa.fields
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
idx = agent.idx[3]
for _ in range(3):
    ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                            f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                            f_logp=f["aux_old_logp"], idx=idx, stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                            diag=None, slab_stride=m.params.P, M=idx.numel(), n_envs=n, T=256, D=4, A=2, clip_range=0.2,
                            vf_coef=0.25, ent_coef=0.01, dbg=dbg)
    torch.cuda.synchronize()
d = dbg.tolist(); k = d[15]
names = ["loads+L0", "mid fwd", "heads+loss", "small grads", "dW(mid)", "dH(mid)", "g1", "first-bwd"] if k == 9 else ["setup", "forward", "loss", "heads-bwd", "dW(mid)", "dH+db(mid)", "first-bwd"]
print("phase cycles:", {names[i] if i < len(names) else i: d[i + 1] - d[i] for i in range(k - 1)}, "total", d[k - 1] - d[0],
      "= %.1f us at 2.4 GHz" % ((d[k - 1] - d[0]) / 2400.0))

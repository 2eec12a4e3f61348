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
lr._derived_layouts()            # params_t / cache_image: the role-split learner does not keep them, the direct calls below need them
m, f = lr.model, mem.soa.fields
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
idx = agent.idx[3]
for _ in range(3):
    ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                            f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                            f_logp=f["aux_old_logp"], idx=idx, stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                            diag=None, slab_stride=m.params.P, M=idx.numel(), n_envs=n, T=256, D=4, A=2, clip_range=0.2,
                            vf_coef=0.25, ent_coef=0.01, dbg=dbg, frag_image=lr.frag, f_packed=lr.packed)
    torch.cuda.synchronize()
d = dbg.tolist()
names = ["start", "h1 ready", "h2 ready", "heads+loss done", "small grads done", "dW done", "dH partials", "g1 ready", "end", "rows gathered"]
print("stamps (cycles from kernel start; needs a -DXRL_TILE_PROBE build):", dict(zip(names, d[:10])))

def mb(dbg_=None):
    ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                            f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                            f_logp=f["aux_old_logp"], idx=idx, stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                            diag=None, slab_stride=m.params.P, M=idx.numel(), n_envs=n, T=256, D=4, A=2, clip_range=0.2,
                            vf_coef=0.25, ent_coef=0.01, frag_image=lr.frag, f_packed=lr.packed,
                            f_rows=lr.rows[3 * 8192 * 8:4 * 8192 * 8] if os.environ.get("ROWS", "1") == "1" else None)


def timed(fn, reps=50):
    g = ops.Graph()
    with g:
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.launch(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


opt = lr.optimizer
print("fused minibatch kernel (specialised) : %.2f us per launch" % timed(mb))
ops.set_fast_kernels(False)
print("fused minibatch kernel (any-shape)   : %.2f us per launch" % timed(mb))
ops.set_fast_kernels(True)
print("grad_reduce (256 slabs)              : %.2f us" % timed(lambda: ops.grad_reduce(lr.fslabs, lr.n_tiles, m.params.P, m.params.P, opt.grad, lr.sumsq)))
print("adam + 4 mirrors                     : %.2f us" % timed(lr.finish_step))

for Msub in (256, 1024, 2048, 4096, 8192):
    def mbs():
        ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                                f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                                f_logp=f["aux_old_logp"], idx=idx[:Msub], stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                                diag=None, slab_stride=m.params.P, M=Msub, n_envs=n, T=256, D=4, A=2, clip_range=0.2,
                                vf_coef=0.25, ent_coef=0.01, frag_image=lr.frag, f_packed=lr.packed)
    print("   specialised kernel with M = %5d rows (%3d workgroups): %.2f us" % (Msub, Msub // 32, timed(mbs)))

clip = lr.grad_clip_norm if lr.use_grad_clip else 0.0
print("reduce + adam fused (1 launch)       : %.2f us" % timed(lambda: ops.reduce_adam(lr.fslabs, lr.n_tiles, m.params.P, m.params.flat, opt.grad, opt.m, opt.v, m.params.P, opt.state, lr.sumsq, clip, lr._mirrors, lr.opt_sync)))

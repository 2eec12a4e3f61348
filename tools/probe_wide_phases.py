"""Shader-clock stamps of xrl_ppo_wide_minibatch (csrc/ppo_wide.hip) at the C4 minibatch (4 096 rows = 256 workgroups):
cycles between the phases of the last tile's actor and critic workgroup, and event-timed launches of the kernel and of the
optimiser launch that follows it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import SyntheticMujocoVecEnv

n, T = 128, 256
cfg = Namespace(agent="PPO", representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", seed=1, parallels=n,
                running_steps=10 ** 9, horizon_size=T, n_epochs=16, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25,
                ent_coef=0.0, clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                distributed_training=False, device="cuda", model_dir="/tmp/x", use_hip_graph=True)
torch.manual_seed(0)
agent = PPO_Agent(cfg, SyntheticMujocoVecEnv(n, seed=4))
agent.rollout(); agent.update()
torch.cuda.synchronize()
lr, m, opt = agent.learner, agent.model, agent.learner.optimizer
bs, P = agent.batch_size, m.params.P
st = {k: v[3 * bs:4 * bs] for k, v in lr._wstage.items()}
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
names = ["start", "loads->LDS", "layer0", "layer1", "heads+loss", "small grads", "dW1", "dH1", "dW0"]


def launch(role=None):
    lr._wide.launch(bs, st["observations"], st["actions"], st["returns"], st["advantages"], st["aux_old_logp"], lr.fslabs, P,
                    lr.fpartials, lr.clip_range, lr.vf_coef, lr.ent_coef, stats=lr.stats[3],
                    dbg=None if role is None else dbg, dbg_role=role or 0)


for role in (0, 1):
    for _ in range(3):
        launch(role)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    print("role", role, "total", int(d[8] - d[0]), "cycles:", ", ".join(f"{names[i]} {int(d[i] - d[i - 1])}" for i in range(1, 9)))


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


clip = lr.grad_clip_norm
ra = lambda: ops.reduce_adam(lr.fslabs, lr.n_tiles, P, m.params.flat, opt.grad, opt.m, opt.v, P, opt.state, lr.sumsq, clip,
                             lr._mirrors, lr.opt_sync)
print("wide kernel        : %.2f us" % timed(lambda: launch()))
print("reduce + adam      : %.2f us" % timed(ra))
print("both, back to back : %.2f us" % timed(lambda: (launch(), ra())))

# ---- acting launch: stamps of the four workgroups of (actor tile 0) inside the captured rollout
agent._wact.act_dbg = torch.zeros(32, dtype=torch.int64, device="cuda")
agent._rollout_graph = None
agent.rollout(); agent.rollout()
torch.cuda.synchronize()
d = agent._wact.act_dbg.cpu().numpy().reshape(4, 8)
an = ["start", "rows + statistics -> LDS", "layer 0", "layer 1 slice", "head slice -> ticket", "combine + sample"]
for part in range(4):
    print("acting, part", part, ":", ", ".join(f"{an[i]} {int(d[part, i] - d[part, i - 1])}" for i in range(1, 6) if d[part, i] > 0),
          "| total", int(max(d[part]) - d[part, 0]))

"""C4 shapes (Gaussian 17-256-256-6, minibatch 4 096): HIP-event times of the pieces of one minibatch update -- the minibatch launch with
the middle layers' weight gradient inside (per-tile partials) and without (rows of g2 / h1 stored for xrl_wide_dw1), the xrl_wide_dw1
launch, and the optimiser launch over 128 slab rows / over 32 parts for the w1 ranges."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
from argparse import Namespace
from xuance_amd import ops
from xuance_amd.ops import call, stream_ptr
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
for _ in range(2):
    agent.rollout(); agent.update()
torch.cuda.synchronize()
lr, m = agent.learner, agent.model
opt, P, M = lr.optimizer, m.params.P, 4096
st = lr._wstage
args = (M, st["observations"][:M], st["actions"][:M], st["returns"][:M], st["advantages"][:M], st["aux_old_logp"][:M], lr.fslabs, P,
        lr.fpartials, 0.2, 0.25, 0.0)
w = lr._wide
out = {}
out["minibatch launch, dW1 inside (us)"] = round(bench._event_time_us(lambda: w.launch(*args, split_dw1=False), 30), 2)
w.launch(*args, split_dw1=True)
d = w.desc
parts = C.c_int32(0)
out["minibatch launch, rows stored (us)"] = round(bench._event_time_us(lambda: call("xrl_ppo_wide_minibatch", C.byref(d), stream_ptr()), 30), 2)
out["xrl_wide_dw1 (us)"] = round(bench._event_time_us(lambda: call("xrl_wide_dw1", C.byref(d), C.byref(parts), stream_ptr()), 30), 2)
out["parts"] = parts.value
red = lambda alt: ops.reduce_adam(lr.fslabs, 128, P, m.params.flat, opt.grad, opt.m, opt.v, P, opt.state, lr.sumsq, 0.5, lr._mirrors, lr.opt_sync, alt=alt)
out["optimiser launch, 128 rows (us)"] = round(bench._event_time_us(lambda: red(None), 30), 2)
out["optimiser launch, w1 ranges over %d parts (us)" % parts.value] = round(bench._event_time_us(lambda: red((parts.value, w.w1_ranges())), 30), 2)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wide_update_pieces.json"), "w"), indent=1)

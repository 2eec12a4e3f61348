"""PPO at the shapes of BASELINE config C4 (HalfCheetah: obs 17, Box(6), Gaussian actor 17-256-256-6, critic
17-256-256-1, 128 envs per GPU, horizon 256, 16 epochs x 8 minibatches of 4096) on one MI355X: rollout = two launches per
vector step (xrl_wide_act_step + the MuJoCo-shaped synthetic provider), update = xrl_ppo_wide_minibatch + the optimiser
launch per minibatch, both phases replayed from hipGraphs (csrc/ppo_wide.hip; config keys use_fused_update / use_fused_acting
= False select the general layered path instead).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
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
R = 5
t0 = time.perf_counter()
for _ in range(R):
    agent.rollout()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(R):
    info = agent.update()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps({"config": "C4 PPO HalfCheetah shapes, 128 envs/GPU x horizon 256, 16 epochs x 8 minibatches of 4096, "
                            "Gaussian 17-256-256-6 + critic 17-256-256-1 (142 605 parameters)",
                  "rollout_ms": round((t1 - t0) / R * 1e3, 3), "update_ms": round((t2 - t1) / R * 1e3, 3),
                  "env_steps_per_s": round(n * T * R / (t2 - t0), 1), "update_us_per_minibatch": round((t2 - t1) / R / 128 * 1e6, 1),
                  "last_info": {k: round(float(v), 5) for k, v in info.items()}}))

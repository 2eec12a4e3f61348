"""rocprofv3 --kernel-trace --stats target: rollouts of PPO on the Atari shape (8 envs x 128 steps) with the acting tail as one launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_secondary as bs
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import SyntheticAtariVecEnv
from argparse import Namespace
n, T = 8, 128
cfg = Namespace(agent="PPO", representation="AC_CNN_Atari", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                fc_hidden_sizes=[512], actor_hidden_size=[], critic_hidden_size=[], activation="relu", seed=1, parallels=n,
                running_steps=10 ** 7, horizon_size=T, n_epochs=4, n_minibatch=4, learning_rate=2.5e-4, vf_coef=0.25, ent_coef=0.01,
                clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=0.5,
                use_obsnorm=False, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5, distributed_training=False, device="cuda",
                model_dir="/tmp/x", use_hip_graph=True)
torch.manual_seed(0)
agent = PPO_Agent(cfg, SyntheticAtariVecEnv(n, seed=5))
for _ in range(6):
    agent.rollout()
torch.cuda.synchronize()

"""Kernel table of ONE recurrent-QMIX update phase (8 updates, 3m.yaml shapes) for rocprofv3 --kernel-trace --stats:
fills the episode ring with one run_episodes pass, then replays the captured update graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv

backprop = len(sys.argv) > 1 and sys.argv[1] == "bptt"
n = 64
cfg = Namespace(q_hidden_size=[64], fc_hidden_sizes=[64], recurrent_hidden_size=64, hidden_dim_mixing_net=32,
                hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=n, running_steps=10 ** 7, buffer_size=5000,
                batch_size=32, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
                decay_step_greedy=50000, sync_frequency=200, training_frequency=1, start_training=0, n_epochs=8,
                use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=True,
                rnn_backprop_agents=backprop, episode_length=60, distributed_training=False, device="cuda", model_dir="/tmp/x")
torch.manual_seed(0)
agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(n, seed=3))
agent.run_episodes(n)
lr, mem = agent.learner, agent.memory
lr.update_from_buffer(mem, 8, seed=1)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    lr._buf_graph.launch()
torch.cuda.synchronize()
print("update_us", (time.perf_counter() - t0) / 50 / 8 * 1e6)

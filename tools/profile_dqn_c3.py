"""Kernel table of the C3 (DQN, Atari shapes, batch 32) update graph for rocprofv3 --kernel-trace --stats, at the configuration's
replay size: a 499 968-transition uint8 ring (28.2 GB), filled on the device before the measured launches (XRL_C3_RING overrides)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from xuance_amd.agents import DQN_Agent
from xuance_amd.envs import SyntheticAtariVecEnv

n = 64
cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 7, buffer_size=int(os.environ.get("XRL_C3_RING", 499968)),
                batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05, decay_step_greedy=10 ** 6,
                sync_frequency=500, training_frequency=n, start_training=10 ** 9, use_grad_clip=False, grad_clip_norm=0.5,
                use_obsnorm=False, use_rewnorm=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
torch.manual_seed(0)
agent = DQN_Agent(cfg, SyntheticAtariVecEnv(n, seed=2))
agent.memory.fill_synthetic(seed=4)
agent.train(40)
lr, mem = agent.learner, agent.memory
assert mem.size == mem.n_size
lr.update_from_buffer(mem, 1, seed=1)
lr.update_from_buffer(mem, 1, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    lr._buf_graph.launch()
torch.cuda.synchronize()
print("update_us", (time.perf_counter() - t0) / 50 * 1e6)

"""Throughput of the off-policy loops at the shapes of BASELINE configs C3 (DQN, Atari frames, CNN) and C5 (QMIX,
3m shapes, feed-forward) on one MI355X: env-steps/s of the whole agent loop and the time of one update
(sample + forward/TD/backward + optimiser).  One JSON line per config; run under rocprofv3 for the kernel table."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np
import torch


def timed_train(agent, warm, steps):
    agent.train(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def update_us(agent, reps=30):
    lr, mem = agent.learner, agent.memory
    for _ in range(3):
        lr.update(**mem.sample()) if not hasattr(agent, "agent_keys") else lr.update(mem.sample())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lr.update(**mem.sample()) if not hasattr(agent, "agent_keys") else lr.update(mem.sample())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def dqn_c3(steps):
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n = 64
    cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                    q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 7,
                    buffer_size=n * 512, batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05,
                    decay_step_greedy=10 ** 6, sync_frequency=500, training_frequency=n, start_training=n * 8,
                    use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/x")
    agent = DQN_Agent(cfg, SyntheticAtariVecEnv(n, seed=2))
    dt = timed_train(agent, 16, steps)
    lr, mem = agent.learner, agent.memory
    lr.update_from_buffer(mem, 1, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        lr._buf_graph.launch()
    torch.cuda.synchronize(); graph_us = (time.perf_counter() - t0) / 30 * 1e6
    return {"update_graph_us": round(graph_us, 1), "config": "C3 DQN, 64 envs x 84x84x4 uint8 frames, CNN 32/64/64 + 512, batch 32, one update per vector step",
            "env_steps_per_s": round(n * steps / dt, 1), "vector_step_us": round(dt / steps * 1e6, 1),
            "update_us": round(update_us(agent), 1), "replay_bytes_per_transition": 2 * 28224 + 12}


def qmix_c5(steps):
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    n = 64
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32,
                    activation="relu", seed=1, parallels=n, running_steps=10 ** 7, buffer_size=n * 78, batch_size=32,
                    learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
                    decay_step_greedy=50000, sync_frequency=200, training_frequency=1, start_training=640, n_epochs=8,
                    use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True,
                    use_rnn=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(n, seed=3))
    dt = timed_train(agent, 20, steps)
    lr, mem = agent.learner, agent.memory
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        lr.update_from_buffer(mem, 8, seed=1)
    torch.cuda.synchronize(); phase_us = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        lr._buf_graph.launch()
    torch.cuda.synchronize(); graph_us = (time.perf_counter() - t0) / 20 * 1e6
    return {"update_phase_us_8_updates": round(phase_us, 1), "update_graph_us_8_updates": round(graph_us, 1), "config": "C5 QMIX (feed-forward), 64 envs/GPU x 3 agents, obs 30 / state 48 / 9 masked actions, batch 32, 8 updates per vector step",
            "env_steps_per_s": round(n * steps / dt, 1), "vector_step_us": round(dt / steps * 1e6, 1),
            "update_us": round(update_us(agent), 1)}


def qmix_c5_rnn(steps, backprop):
    """configs/qmix/sc2/3m.yaml defaults: Basic_RNN (fc 64 + GRU 64), 60-step episodes, batch 32 EPISODES (1 920 steps,
    5 856 agent rows), 8 updates after every n_envs episodes.  backprop: False = the reference's detached agents."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    n = 64
    cfg = Namespace(q_hidden_size=[64], fc_hidden_sizes=[64], recurrent_hidden_size=64, hidden_dim_mixing_net=32,
                    hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=n, running_steps=10 ** 7, buffer_size=5000,
                    batch_size=32, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
                    decay_step_greedy=50000, sync_frequency=200, training_frequency=1, start_training=1000, n_epochs=8,
                    use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True,
                    use_rnn=True, rnn_backprop_agents=backprop, episode_length=60, distributed_training=False, device="cuda",
                    model_dir="/tmp/x")
    agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(n, seed=3))
    agent.train(60)
    torch.cuda.synchronize()
    s0 = agent.current_step; t0 = time.perf_counter()
    agent.train(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env_steps = agent.current_step - s0
    lr, mem = agent.learner, agent.memory
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        lr._buf_graph.launch()
    torch.cuda.synchronize(); graph_us = (time.perf_counter() - t0) / 20 * 1e6
    return {"config": "C5 QMIX recurrent (3m.yaml: fc 64 + GRU 64, 60-step episodes), 64 envs/GPU x 3 agents, batch 32 episodes, "
                      "8 updates per 64 episodes, agents " + ("trained (BPTT)" if backprop else "detached (reference behaviour)"),
            "env_steps_per_s": round(env_steps / dt, 1), "update_graph_us_8_updates": round(graph_us, 1),
            "update_us": round(graph_us / 8, 1), "transitions_per_update": 32 * 60}


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.manual_seed(0); np.random.seed(0)
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    if which in ("all", "ff"):
        for fn in (dqn_c3, qmix_c5):
            print(json.dumps(fn(steps)), flush=True)
    if which in ("all", "rnn"):
        for bp in (False, True):
            print(json.dumps(qmix_c5_rnn(steps, bp)), flush=True)

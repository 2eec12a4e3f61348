"""Per-call times of the C4-shape PPO loop (tools/bench_c4.py's agent): every rollout / update call by itself, whether the update
graph was captured again between calls, and the captured graph's launch by itself."""
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


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 3)


out = {"rollout_ms": [], "update_ms": [], "graph_ids": [], "ws_sig": []}
for _ in range(6):
    out["rollout_ms"].append(timed(agent.rollout))
    out["update_ms"].append(timed(agent.update))
    out["graph_ids"].append(id(agent._update_graph))
    out["ws_sig"].append(repr(agent._ws_sig()))
out["updates_back_to_back_ms"] = [timed(agent.update) for _ in range(4)]
out["graph_launch_ms"] = [timed(agent._update_graph.launch) for _ in range(4)]
lr = agent.learner
out["wide"] = getattr(lr, "_wide", None) is not None
out["fused_optimizer"] = bool(getattr(lr, "_fused_optimizer_ok", lambda: None)()) if callable(getattr(lr, "_fused_optimizer_ok", None)) else None
print(json.dumps(out))

"""C4 shapes (128 envs x horizon 256, Gaussian 17-256-256-6): rollout phase with the whole-rollout launch (csrc/rollout_wide.hip) and with
the launches per vector step, HIP-event times of the pieces, shader-clock stamps of workgroup 0 at step T / 2."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import torch
import bench
from argparse import Namespace
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import SyntheticMujocoVecEnv

n, T = 128, 256
out = {}
for whole in (True, False):
    cfg = Namespace(agent="PPO", representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                    critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", seed=1, parallels=n,
                    running_steps=10 ** 9, horizon_size=T, n_epochs=16, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25,
                    ent_coef=0.0, clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                    grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                    distributed_training=False, device="cuda", model_dir="/tmp/x", use_hip_graph=True, use_wide_rollout=whole)
    torch.manual_seed(0)
    agent = PPO_Agent(cfg, SyntheticMujocoVecEnv(n, seed=4))
    for _ in range(2):
        agent.rollout(); agent.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        agent.rollout()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3):
        agent.update()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = {"rollout_ms": round((t1 - t0) / 5 * 1e3, 3), "update_ms": round((t2 - t1) / 3 * 1e3, 3)}
    if whole:
        wr = agent._wide_rollout()
        assert wr is not None
        r["status"] = agent._wr_status.tolist()
        r["actor_launch_us"] = round(bench._event_time_us(lambda: wr.run(0, T), 5), 1)
        M, A, f = n * T, 6, agent.memory.soa.fields
        r["values_pass_us"] = round(bench._event_time_us(lambda: (agent.model.forward(f["observations"].view(M, -1), M)), 5), 1)
        fl = int(os.environ.get("WIDE_FLAGS", "0"), 0)
        dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
        wr.run(0, T, flags=fl, dbg=dbg); torch.cuda.synchronize()
        d = dbg.tolist()
        r["waves: prepare done | messages in | merge done (cycles from the step's start), poll rounds"] = [
            [d[16 + w] - d[0], d[24 + w] - d[0], d[40 + w] - d[0], d[32 + w]] for w in range(8)]
        names = ["statistics (poll + merge)", "barrier 1 + first layer + barrier 2", "middle layer + mean partials + barrier 3",
                 "sample + simulator noise + barrier 4", "dynamics + records + barriers 5a / 5", "partial sums + message"]
        r["phase_cycles_of_step_128"] = {names[i]: d[i + 1] - d[i] for i in range(6)}
        r["us_per_step"] = round(r["actor_launch_us"] / T, 3)
    out["whole" if whole else "per_step"] = r
    print(json.dumps(r))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wide_rollout.json"), "w"), indent=1)

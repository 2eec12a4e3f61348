"""Scratch: the general on-policy path at the reference's DEFAULT sizes (configs/ppo/classic_control/*.yaml: parallels 10, horizon 256):
rollout time per vector step with the acting pass as one launch (xrl_trunk_forward16, config.use_trunk_forward) and with the layered
forward + xrl_policy_sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import xuance_amd.envs as envs
from xuance_amd.agents import PPO_Agent

for name, n in (("DeviceAcrobotVecEnv", 10), ("DeviceMountainCarVecEnv", 10), ("DevicePendulumVecEnv", 10), ("DeviceAcrobotVecEnv", 64)):
    row = []
    for tf in (False, True):
        cfg = bench.make_config(n, 256, 1, 0)
        cfg.use_trunk_forward = tf
        if name == "DevicePendulumVecEnv":
            cfg.activation_action = "tanh"
        torch.manual_seed(1)
        agent = PPO_Agent(cfg, getattr(envs, name)(n, seed=1))
        for _ in range(3):
            agent.rollout(); agent.update()
        torch.cuda.synchronize()
        us = bench._event_time_us(agent.rollout, 10)
        assert (agent._trunk_forward() is not None) == tf
        row.append(us / 256)
    print("%-24s %3d envs: %.1f us per vector step layered -> %.1f us with the one-launch acting pass" % (name, n, row[0], row[1]), flush=True)

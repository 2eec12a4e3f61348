"""A/B of the reference's stored-state rule (QMIX_Agents.reference_state_broadcast) in the two QMIX loops on the same box:
env-steps/s of tools/bench_secondary.py's qmix_3m lines with the rule off / on (and, for the captured recurrent loop, as a launch of
its own vs riding in the gate launch), alternating."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_secondary as bs


def rate(rnn, **over):
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    cfg = bs._qmix_cfg(64, rnn)
    for k, v in over.items():
        setattr(cfg, k, v)
    agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(64, seed=3))
    agent.train(60 if rnn else 20)
    torch.cuda.synchronize()
    s0, t0 = agent.current_step, time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        agent.train(180 if rnn else 200)
        torch.cuda.synchronize()
    return round((agent.current_step - s0) / (time.perf_counter() - t0), 1)


out = {"ff": {}, "gru": {}}
for rep in range(3):
    for name, kw in (("off", dict(reference_state_broadcast=False)), ("on", dict(reference_state_broadcast=True))):
        out["ff"].setdefault(name, []).append(rate(False, **kw))
    for name, kw in (("off", dict(reference_state_broadcast=False)), ("own launch", dict(reference_state_broadcast=True, state_broadcast_in_gate=False)),
                     ("in gate", dict(reference_state_broadcast=True, state_broadcast_in_gate=True))):
        out["gru"].setdefault(name, []).append(rate(True, **kw))
print(json.dumps(out))

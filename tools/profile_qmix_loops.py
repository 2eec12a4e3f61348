"""The two QMIX-3m agent loops of bench.py's `secondary` (feed-forward and GRU agents, 64 envs) for rocprofv3 passes:
`python tools/profile_qmix_loops.py ff|gru` runs a warm-up and a short timed stretch; see tools/collect_pmc_qmix.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import bench_secondary as bs
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv

rnn = sys.argv[1] == "gru"
agent = QMIX_Agents(bs._qmix_cfg(64, rnn), SyntheticSMACVecEnv(64, seed=3))
agent.train(60 if rnn else 20)
torch.cuda.synchronize()
agent.train(120 if rnn else 40)
torch.cuda.synchronize()

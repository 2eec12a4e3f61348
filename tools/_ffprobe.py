import sys
sys.path.insert(0, "/root/repo")
from tools import bench_secondary as bs
import torch
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv
agent = QMIX_Agents(bs._qmix_cfg(64, False), SyntheticSMACVecEnv(64, seed=3))
agent.train(30)
torch.cuda.synchronize()
agent.train(100)
torch.cuda.synchronize()

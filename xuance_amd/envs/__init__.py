from .cartpole import DeviceCartPoleVecEnv
from .host_cartpole import NumpyCartPoleEnv
from .shm_vec_env import ShmSubprocVecEnv
from .shm_vec_maenv import ShmSubprocVecMultiAgentEnv
from .dummy_vec_env import DummyVecEnv, DummyVecMultiAgentEnv, HostSMACLikeEnv
from .synthetic import SyntheticAtariVecEnv, SyntheticMujocoVecEnv, SyntheticSMACVecEnv

REGISTRY_VEC_ENV = {"DeviceCartPoleVecEnv": DeviceCartPoleVecEnv, "SyntheticAtariVecEnv": SyntheticAtariVecEnv,
                    "SyntheticMujocoVecEnv": SyntheticMujocoVecEnv, "SyntheticSMACVecEnv": SyntheticSMACVecEnv, "ShmSubprocVecEnv": ShmSubprocVecEnv,
                    "ShmSubprocVecMultiAgentEnv": ShmSubprocVecMultiAgentEnv,
                    "DummyVecEnv": DummyVecEnv, "DummyVecMultiAgentEnv": DummyVecMultiAgentEnv}

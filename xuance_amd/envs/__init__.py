from .cartpole import DeviceCartPoleVecEnv
from .host_cartpole import NumpyCartPoleEnv
from .shm_vec_env import ShmSubprocVecEnv
from .synthetic import SyntheticAtariVecEnv, SyntheticMujocoVecEnv, SyntheticSMACVecEnv

REGISTRY_VEC_ENV = {"DeviceCartPoleVecEnv": DeviceCartPoleVecEnv, "SyntheticAtariVecEnv": SyntheticAtariVecEnv,
                    "SyntheticMujocoVecEnv": SyntheticMujocoVecEnv, "SyntheticSMACVecEnv": SyntheticSMACVecEnv, "ShmSubprocVecEnv": ShmSubprocVecEnv}

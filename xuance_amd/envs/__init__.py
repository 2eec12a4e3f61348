from .cartpole import DeviceCartPoleVecEnv
from .synthetic import SyntheticAtariVecEnv, SyntheticMujocoVecEnv, SyntheticSMACVecEnv

REGISTRY_VEC_ENV = {"DeviceCartPoleVecEnv": DeviceCartPoleVecEnv, "SyntheticAtariVecEnv": SyntheticAtariVecEnv,
                    "SyntheticMujocoVecEnv": SyntheticMujocoVecEnv, "SyntheticSMACVecEnv": SyntheticSMACVecEnv}

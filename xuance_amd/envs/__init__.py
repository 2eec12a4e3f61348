from .cartpole import DeviceCartPoleVecEnv
from .classic import DevicePendulumVecEnv, DeviceMountainCarVecEnv, DeviceAcrobotVecEnv
from .host_cartpole import NumpyCartPoleEnv
from .shm_vec_env import ShmSubprocVecEnv
from .shm_vec_maenv import ShmSubprocVecMultiAgentEnv
from .dummy_vec_env import DummyVecEnv, DummyVecMultiAgentEnv, HostSMACLikeEnv
from .synthetic import SyntheticAtariVecEnv, SyntheticMujocoVecEnv, SyntheticSMACVecEnv
from .recorded import RecordedVecEnv, RecordedMultiAgentVecEnv, TapeCartPoleVecEnv, TapeControlVecEnv

REGISTRY_VEC_ENV = {"DeviceCartPoleVecEnv": DeviceCartPoleVecEnv, "DevicePendulumVecEnv": DevicePendulumVecEnv,
                    "DeviceMountainCarVecEnv": DeviceMountainCarVecEnv, "DeviceAcrobotVecEnv": DeviceAcrobotVecEnv, "SyntheticAtariVecEnv": SyntheticAtariVecEnv,
                    "SyntheticMujocoVecEnv": SyntheticMujocoVecEnv, "SyntheticSMACVecEnv": SyntheticSMACVecEnv, "ShmSubprocVecEnv": ShmSubprocVecEnv,
                    "ShmSubprocVecMultiAgentEnv": ShmSubprocVecMultiAgentEnv,
                    "DummyVecEnv": DummyVecEnv, "DummyVecMultiAgentEnv": DummyVecMultiAgentEnv, "RecordedVecEnv": RecordedVecEnv,
                    "RecordedMultiAgentVecEnv": RecordedMultiAgentVecEnv}

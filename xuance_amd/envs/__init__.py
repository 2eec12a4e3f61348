from .cartpole import DeviceCartPoleVecEnv

REGISTRY_VEC_ENV = {"DeviceCartPoleVecEnv": DeviceCartPoleVecEnv}

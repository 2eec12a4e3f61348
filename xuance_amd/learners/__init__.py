from .base import Learner, AdamHandle, LinearLRHandle
from .ppo_learner import PPO_Learner

PPOCLIP_Learner = PPO_Learner   # the north-star's name for the same class (registry key is "PPO_Learner")

REGISTRY_Learners = {"PPO_Learner": PPO_Learner, "PPOCLIP_Learner": PPO_Learner}

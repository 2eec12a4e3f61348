from .base import Learner, AdamHandle, LinearLRHandle
from .ppo_learner import PPO_Learner, A2C_Learner, PG_Learner, PPOKL_Learner
from .dqn_learner import DQN_Learner, DDQN_Learner, DuelDQN_Learner, PerDQN_Learner
from .qmix_learner import QMIX_Learner, VDN_Learner, IQL_Learner

PPOCLIP_Learner = PPO_Learner   # the north-star's name for the same class (registry key is "PPO_Learner")

REGISTRY_Learners = {"PPO_Learner": PPO_Learner, "A2C_Learner": A2C_Learner, "PG_Learner": PG_Learner, "PPOKL_Learner": PPOKL_Learner, "PPOCLIP_Learner": PPO_Learner, "DQN_Learner": DQN_Learner,
                     "DDQN_Learner": DDQN_Learner, "DuelDQN_Learner": DuelDQN_Learner, "PerDQN_Learner": PerDQN_Learner, "QMIX_Learner": QMIX_Learner, "VDN_Learner": VDN_Learner,
                     "IQL_Learner": IQL_Learner}

from .ppo_agent import PPO_Agent, A2C_Agent, PG_Agent, PPOKL_Agent
from .dqn_agent import DQN_Agent, DDQN_Agent, DuelDQN_Agent, PerDQN_Agent
from .qmix_agents import QMIX_Agents, VDN_Agents, IQL_Agents

REGISTRY_Agents = {"PPO": PPO_Agent, "A2C": A2C_Agent, "PG": PG_Agent, "PPOKL": PPOKL_Agent, "DQN": DQN_Agent, "DDQN": DDQN_Agent, "Duel_DQN": DuelDQN_Agent, "PerDQN": PerDQN_Agent, "QMIX": QMIX_Agents, "VDN": VDN_Agents, "IQL": IQL_Agents}

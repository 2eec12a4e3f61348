from .ppo_agent import PPO_Agent

REGISTRY_Agents = {"PPO": PPO_Agent}

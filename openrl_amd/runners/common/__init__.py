from ...modules.common.ppo_net import PPONet  # BASELINE.json names it under runners.common as well
from .ppo_agent import PPOAgent

__all__ = ["PPOAgent", "PPONet"]

from ...modules.common.ppo_net import PPONet  # BASELINE.json names it under runners.common as well
from .a2c_agent import A2CAgent
from .mat_agent import MATAgent
from .ppo_agent import PPOAgent

__all__ = ["A2CAgent", "MATAgent", "PPOAgent", "PPONet"]

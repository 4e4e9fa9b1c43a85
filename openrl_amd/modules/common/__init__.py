from .ppo_net import PPONet

__all__ = ["PPONet"]

"""openrl_amd - MI355X-native on-policy rollout + PPO update engine behind OpenRL's drop-in API.

    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

See DESIGN.md for the hot path, its boundary (``include/orl_hip.h``) and the kernels.
"""
__version__ = "0.1.0"

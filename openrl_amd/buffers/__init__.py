from .normal_buffer import NormalReplayBuffer
from .replay_data import ReplayData

__all__ = ["NormalReplayBuffer", "ReplayData"]

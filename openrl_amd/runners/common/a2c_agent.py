"""``A2CAgent`` (``openrl/runners/common/a2c_agent.py:34-76``): ``PPOAgent`` whose ``train`` defaults to
``A2CAlgorithm``."""
from __future__ import annotations

from typing import Optional, Type

from ...algorithms.a2c import A2CAlgorithm
from ...drivers.onpolicy_driver import OnPolicyDriver as Driver
from .ppo_agent import PPOAgent


class A2CAgent(PPOAgent):
    def train(self, total_time_steps: int, callback=None, train_algo_class: Type = A2CAlgorithm, logger=None,
              driver_class: Type = Driver) -> None:
        super().train(total_time_steps, callback, train_algo_class, logger, driver_class)

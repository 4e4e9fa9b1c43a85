"""``MATAgent`` (``openrl/runners/common/mat_agent.py:27-49``): ``PPOAgent`` whose ``train`` defaults to
``MATAlgorithm`` and logs under the project name "MATAgent"."""
from __future__ import annotations

from typing import Type

from ...algorithms.mat import MATAlgorithm
from ...drivers.onpolicy_driver import OnPolicyDriver as Driver
from ...utils.logger import Logger
from .ppo_agent import PPOAgent


class MATAgent(PPOAgent):
    def train(self, total_time_steps: int, callback=None, train_algo_class: Type = MATAlgorithm, logger=None,
              driver_class: Type = Driver) -> None:
        if logger is None:
            logger = Logger(cfg=self._cfg, project_name="MATAgent", scenario_name=self._env.env_name,
                            exp_name=self.exp_name, log_path=self.run_dir, use_wandb=self._use_wandb,
                            use_tensorboard=self._use_tensorboard)
        super().train(total_time_steps, callback, train_algo_class, logger, driver_class)

"""``A2CAlgorithm`` (``openrl/algorithms/a2c.py:27-145``): the PPO update chain of ``ppo.py`` with the policy loss
``-adv * log_prob`` instead of the clipped surrogate, one minibatch per epoch, and no ``ratio`` in ``train_info``.
Same fused HIP kernels - the loss variant is a flag bit of ``orl_ppo_hparams`` (include/orl_hip.h)."""
from __future__ import annotations

from typing import Union

import torch

from .ppo import PPOAlgorithm


class A2CAlgorithm(PPOAlgorithm):
    def __init__(self, cfg, init_module, agent_num: int = 1, device: Union[str, torch.device] = "cuda:0") -> None:
        super().__init__(cfg, init_module, agent_num, device)
        self.num_mini_batch = 1  # a2c.py:37
        self.hp.reserved |= 2    # policy-gradient loss

    def train(self, buffer, turn_on: bool = True):
        train_info = super().train(buffer, turn_on)
        train_info.pop("ratio", None)  # a2c.py:142-145
        return train_info

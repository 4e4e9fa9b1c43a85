"""``A2CAlgorithm`` (``openrl/algorithms/a2c.py:27-145``): the PPO update chain of ``ppo.py`` with the policy loss
``-adv * log_prob`` instead of the clipped surrogate, one minibatch per epoch, and no ``ratio`` in ``train_info``.
Same fused HIP kernels - the loss variant is a flag bit of ``orl_ppo_hparams`` (include/orl_hip.h)."""
from __future__ import annotations

from typing import Union

import torch

from .ppo import INFO_KEYS, PPOAlgorithm


class A2CAlgorithm(PPOAlgorithm):
    info_keys = INFO_KEYS[:5]  # no "ratio" (a2c.py:142-145); the dict stays lazy - nothing is popped on the host

    def __init__(self, cfg, init_module, agent_num: int = 1, device: Union[str, torch.device] = "cuda:0") -> None:
        super().__init__(cfg, init_module, agent_num, device)
        self.num_mini_batch = 1  # a2c.py:37
        self.hp.reserved |= 2    # policy-gradient loss

"""Hyper-parameter capture shared by the algorithms: the attribute names the reference's algorithms read from ``self``
(``openrl/algorithms/base_algorithm.py:24-86``), filled from the config by table instead of one statement each."""
from __future__ import annotations

import torch

from .. import _native as nat

# cfg.<name> -> self.<name>
_SAME_NAME = ("clip_param", "ppo_epoch", "num_mini_batch", "mini_batch_size", "data_chunk_length", "value_loss_coef",
              "entropy_coef", "max_grad_norm", "huber_delta", "dual_clip_ppo")
# cfg.<flag> -> self._<flag> (the reference's private spellings; two of them are abbreviated there)
_PRIVATE = {"use_recurrent_policy": "_use_recurrent_policy", "use_naive_recurrent_policy": "_use_naive_recurrent",
            "use_max_grad_norm": "_use_max_grad_norm", "use_clipped_value_loss": "_use_clipped_value_loss",
            "use_huber_loss": "_use_huber_loss", "use_popart": "_use_popart", "use_valuenorm": "_use_valuenorm",
            "use_value_active_masks": "_use_value_active_masks", "use_policy_active_masks": "_use_policy_active_masks",
            "use_policy_vhead": "_use_policy_vhead", "use_adv_normalize": "_use_adv_normalize"}


class BaseAlgorithm:
    def __init__(self, cfg, init_module, agent_num: int, device="cuda:0"):
        self.cfg, self.algo_module, self.agent_num = cfg, init_module, agent_num
        self.device = nat.require_gpu(device)  # no CPU path: raises without a HIP device
        self.tpdv = {"dtype": torch.float32, "device": self.device}
        self.world_size = getattr(init_module, "world_size", None) or 1
        for name in _SAME_NAME:  # mini_batch_size is carried but never forwarded by PPO (ppo.py:378-380)
            setattr(self, name, getattr(cfg, name))
        for flag, attr in _PRIVATE.items():
            setattr(self, attr, getattr(cfg, flag))
        self.dual_clip_coeff = float(cfg.dual_clip_coeff)
        if self._use_popart and self._use_valuenorm:
            raise AssertionError("use_popart and use_valuenorm can not both be True")

    def train(self, buffer, turn_on=True):
        raise NotImplementedError

    def _set_mode(self, training: bool) -> None:
        for model in self.algo_module.models.values():
            model.train() if training else model.eval()

    def prep_training(self):
        self._set_mode(True)

    def prep_rollout(self):
        self._set_mode(False)

"""Hyper-parameter capture shared by the algorithms (``openrl/algorithms/base_algorithm.py:24-86``)."""
from __future__ import annotations

import torch

from .. import _native as nat


class BaseAlgorithm:
    def __init__(self, cfg, init_module, agent_num: int, device="cuda:0"):
        self.cfg = cfg
        self.device = nat.require_gpu(device)
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.algo_module = init_module
        self.world_size = self.algo_module.world_size or 1
        self.clip_param = cfg.clip_param
        self.ppo_epoch = cfg.ppo_epoch
        self.num_mini_batch = cfg.num_mini_batch
        self.mini_batch_size = cfg.mini_batch_size  # carried, never forwarded by PPO (ppo.py:378-380)
        self.data_chunk_length = cfg.data_chunk_length
        self.value_loss_coef = cfg.value_loss_coef
        self.entropy_coef = cfg.entropy_coef
        self.max_grad_norm = cfg.max_grad_norm
        self.huber_delta = cfg.huber_delta
        self._use_recurrent_policy = cfg.use_recurrent_policy
        self._use_naive_recurrent = cfg.use_naive_recurrent_policy
        self._use_max_grad_norm = cfg.use_max_grad_norm
        self._use_clipped_value_loss = cfg.use_clipped_value_loss
        self._use_huber_loss = cfg.use_huber_loss
        self._use_popart = cfg.use_popart
        self._use_valuenorm = cfg.use_valuenorm
        self._use_value_active_masks = cfg.use_value_active_masks
        self._use_policy_active_masks = cfg.use_policy_active_masks
        self._use_policy_vhead = cfg.use_policy_vhead
        self.agent_num = agent_num
        self._use_adv_normalize = cfg.use_adv_normalize
        self.dual_clip_ppo = cfg.dual_clip_ppo
        self.dual_clip_coeff = float(cfg.dual_clip_coeff)
        assert not (self._use_popart and self._use_valuenorm), "use_popart and use_valuenorm can not both be True"

    def train(self, buffer, turn_on=True):
        raise NotImplementedError

    def prep_training(self):
        for model in self.algo_module.models.values():
            model.train()

    def prep_rollout(self):
        for model in self.algo_module.models.values():
            model.eval()

"""``MATAlgorithm`` (``openrl/algorithms/mat.py:21-38``): ``PPOAlgorithm`` with

* ONE summed loss ``policy_loss - entropy_coef * dist_entropy + value_loss_coef * value_loss`` and one backward
  (``construct_loss_list``).  With the separate policy / value towers of ``PPOModule`` no parameter is shared, so the
  summed loss's gradients are, tower by tower, exactly the two losses' gradients of ``PPOAlgorithm`` - the same fused
  kernels run; and
* ``feed_forward_generator_transformer`` (``buffers/replay_data.py:707-804``): a minibatch is a set of (step, env)
  PAIRS drawn with ``torch.randperm(T * N)``, every pair bringing ALL its agents' rows, in agent order
  (``_shuffle_agent_grid`` keeps the agent axis) - instead of ``randperm(T * N * A)`` over single rows.

The reference pairs this algorithm with its transformer network (``MAT_network.py``, a different model family that
is not built here); like the reference's own ``tests/test_algorithm/test_mat_algorithm.py`` it also runs on the MLP
``PPOModule``, which is what this class does."""
from __future__ import annotations

from typing import Union

import torch

from .. import ops
from .ppo import PPOAlgorithm


def mat_group_rows(groups: torch.Tensor, agent_num: int) -> torch.Tensor:
    """Row indices ``(t * N + n) * A + a`` of the agents ``a = 0..A-1`` of each (step, env) pair in ``groups``, pair by
    pair: ``[index, N, dim] -> [index * N, dim]`` of replay_data.py:770-772."""
    a = torch.arange(agent_num, dtype=groups.dtype, device=groups.device)
    return (groups.unsqueeze(1) * agent_num + a).reshape(-1)


class MATAlgorithm(PPOAlgorithm):
    def __init__(self, cfg, init_module, agent_num: int = 1, device: Union[str, torch.device] = "cuda:0") -> None:
        super().__init__(cfg, init_module, agent_num, device)
        if self.recurrent:
            raise NotImplementedError("MATAlgorithm with a recurrent policy is not built (the transformer generator "
                                      "hands stored hidden states to a feed-forward pass)")
        self.fuse_next_perm = False  # the fused next-epoch permutation is over rows, this one is over (step, env) pairs

    def _update_minibatch(self, buffer, idx, mb: int, turn_on: bool, next_perm=None):
        # the reference's MATAlgorithm.construct_loss_list (mat.py:24-38) ignores ``turn_on``: the summed loss always
        # contains policy_loss, so the policy is stepped even when the caller passes turn_on=False
        return super()._update_minibatch(buffer, idx, mb, True, next_perm)

    def _minibatch_indices(self, M: int, perm=None):
        A = int(self.agent_num)
        assert M % A == 0
        G = M // A                       # batch_size = n_rollout_threads * episode_length (replay_data.py:717)
        mbg = G // self.num_mini_batch   # mini_batch_size, in pairs
        n_batches = self.num_mini_batch
        if self.perm_mode == "identity":
            if self.num_mini_batch == 1 and G * A == M:
                return [None], M
            groups = torch.arange(G, dtype=torch.int64, device=self.device)
        elif self.perm_mode == "device":
            n, seed, sid, vn = self._perm_job(G)
            self._vn_in_perm = vn is not None
            groups = ops.perm_feistel(n, seed, sid, self.device, vn)
        else:
            groups = torch.randperm(G).to(self.device, non_blocking=True)  # replay_data.py:733
        return [mat_group_rows(groups[b * mbg:(b + 1) * mbg], A) for b in range(n_batches)], mbg * A

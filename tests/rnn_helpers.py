"""Helpers for the recurrent (GRU) parity tests - CPU and GPU."""
import numpy as np
import torch

from oracle import ppo_oracle as po
from oracle import rnn_oracle as ro
from tests import helpers as H

RNN_CASES = ["train_recurrent", "train_recurrent_chunk5", "train_naive_recurrent"]


def rnn_specs(g):
    Dp, Dc = g["buf_policy_obs"].shape[-1], g["buf_critic_obs"].shape[-1]
    if "buf_action_masks" in g:
        pspec = ro.RnnTowerSpec(Dp, g["buf_action_masks"].shape[-1], po.HEAD_CATEGORICAL)
    else:
        pspec = ro.RnnTowerSpec(Dp, g["buf_actions"].shape[-1], po.HEAD_GAUSSIAN)
    return pspec, ro.RnnTowerSpec(Dc, 1, po.HEAD_VALUE)


def rnn_oracle_replay(g):
    cfg = H.case_cfg(g)
    hp = po.hyper_from_cfg(cfg)
    pspec, cspec = rnn_specs(g)
    ptheta, ctheta = torch.tensor(g["theta_p0"]).clone(), torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    vn = po.ValueNormOracle() if cfg.use_valuenorm else None
    buf = H.case_buffer(g)
    torch.manual_seed(int(g["perm_seed"]))
    # naive_recurrent_generator (replay_data.py:806-960) = whole trajectories per lane: chunks of length T
    L = g["buf_actions"].shape[0] if (cfg.use_naive_recurrent_policy and not cfg.use_recurrent_policy) else cfg.data_chunk_length
    info, adv, used = ro.train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf, cfg.ppo_epoch,
                                   cfg.num_mini_batch, L)
    return dict(ptheta=ptheta.numpy(), ctheta=ctheta.numpy(), info=info, adv=adv, used=used,
                vn=None if vn is None else vn.state(), cfg=cfg, hp=hp, pspec=pspec, cspec=cspec)

"""Shared helpers for the parity tests (CPU and GPU)."""
import os

import numpy as np
import torch

from openrl_amd.configs.config import default_cfg
from oracle import ppo_oracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_CASES = ["train_discrete", "train_discrete_masks", "train_gaussian", "train_novn_proper", "train_popart"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True))


def case_cfg(g, N=None, T=None):
    cfg = default_cfg(str(g["argv"]).split())
    return cfg


def case_specs(g):
    D = g["buf_policy_obs"].shape[-1]
    a_w = g["buf_actions"].shape[-1]
    if "buf_action_masks" in g:
        pspec = po.TowerSpec(D, g["buf_action_masks"].shape[-1], po.HEAD_CATEGORICAL)
    else:
        pspec = po.TowerSpec(D, a_w, po.HEAD_GAUSSIAN)
    cspec = po.TowerSpec(D, 1, po.HEAD_VALUE)
    return pspec, cspec


def case_buffer(g):
    buf = {k[4:]: g[k] for k in g if k.startswith("buf_")}
    buf.setdefault("action_masks", None)
    return buf


def oracle_replay(g):
    """Replay the golden case's PPOAlgorithm.train with the oracle; returns final thetas, info, vn state."""
    cfg = case_cfg(g)
    hp = po.hyper_from_cfg(cfg)
    nmb = cfg.num_mini_batch
    if "a2c" in g:  # A2CAlgorithm: policy-gradient loss, num_mini_batch forced to 1 (a2c.py:37)
        hp.a2c, nmb = True, 1
    pspec, cspec = case_specs(g)
    ptheta = torch.tensor(g["theta_p0"]).clone()
    ctheta = torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    vn = po.ValueNormOracle() if cfg.use_valuenorm else None
    buf = case_buffer(g)
    torch.manual_seed(int(g["perm_seed"]))
    info, adv, used = po.train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf, cfg.ppo_epoch, nmb)
    return dict(ptheta=ptheta.numpy(), ctheta=ctheta.numpy(), info=info, adv=adv, used=used,
                vn=None if vn is None else vn.state(), cfg=cfg, hp=hp, pspec=pspec, cspec=cspec)

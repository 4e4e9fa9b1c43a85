"""Shared helpers for the parity tests (CPU and GPU)."""
import os

import numpy as np
import torch

from openrl_amd.configs.config import default_cfg
from oracle import ppo_oracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def experiments_built() -> bool:
    """True when the loaded liborl_hip.so carries the comparison kernels that lost their A/B (ORL_BUILD_EXPERIMENTS: the
    streamed recurrent row kernel, the ticketed one-launch optimiser step, the fp32-MFMA / two-image tower pairs).  The
    shipped build leaves them out; their tests run once per round against an experimental build
    (``ORL_BUILD_DEFS=-DORL_BUILD_EXPERIMENTS=1 python -m openrl_amd.csrc.build --force``, profiles/rNN_pytest_gpu_experiments.log)."""
    try:
        from openrl_amd import _native

        return bool(_native.load(build_if_missing=False).orl_build_experiments())
    except Exception:
        return False


def experimental(*values):
    """pytest.param(...) that skips unless the experimental build is loaded."""
    import pytest

    return pytest.param(*values, marks=pytest.mark.skipif(not experiments_built(),
                                                          reason="comparison kernel: needs an ORL_BUILD_EXPERIMENTS library"))
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


def oracle_replay(g, skip_epochs=0):
    """Replay the golden case's PPOAlgorithm.train with the oracle; returns final thetas, info, vn state.
    ``skip_epochs`` > 0 drops that many trailing epochs (the negative control of the update-parity bar)."""
    cfg = case_cfg(g)
    cfg.ppo_epoch -= skip_epochs
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


# ---- parity on the weight UPDATE (round-3 VERDICT item 1) ------------------------------------------------------------
# A PPO update moves a parameter by ~lr per Adam step (5e-4 by default), i.e. by 1e-4 .. 5e-3 over a golden case, while
# the weights themselves are O(0.1): `assert_allclose(theta_1, ...)` at rtol 2e-3 / atol 3e-5 passes on most entries
# with NO update at all.  The claim a parity test has to carry is about  d_theta = theta_1 - theta_0 :
#   * cosine(d_gpu, d_ref) >= DTHETA_COS per tower, and
#   * |d_gpu - d_ref| <= DTHETA_REL * |d_ref| + DTHETA_REL * median|d_ref|  on >= DTHETA_FRAC of the entries.
# The allowed exceptions (< 1 %) are Adam's sign amplification: an entry whose gradient is ~0 in some epoch gets
# +-lr from Adam whatever the gradient's magnitude, so fp32 summation-order noise in a near-zero gradient is turned into
# an O(lr) difference.  Reference: openrl/algorithms/ppo.py:383-458 (the update whose result theta_1 is).
DTHETA_COS, DTHETA_REL, DTHETA_FRAC = 0.9999, 0.02, 0.99


def dtheta_stats(theta0, got, ref):
    theta0, got, ref = (np.asarray(a, dtype=np.float64).ravel() for a in (theta0, got, ref))
    dg, dr = got - theta0, ref - theta0
    nr = np.linalg.norm(dr)
    cos = float(dg @ dr / (np.linalg.norm(dg) * nr)) if nr > 0 and np.linalg.norm(dg) > 0 else 0.0
    med = float(np.median(np.abs(dr)))
    ok = np.abs(dg - dr) <= DTHETA_REL * np.abs(dr) + DTHETA_REL * med
    return dict(cos=cos, frac=float(ok.mean()), med=med, n=int(dr.size), worst=float(np.abs(dg - dr).max()),
                rel_l2=float(np.linalg.norm(dg - dr) / nr) if nr > 0 else float("inf"))


# Per-block bar (round-4 VERDICT item 7): the 1 % exception budget above is blind to WHERE the exceptions fall - an update that
# is wrong only in a 2..6-entry block (b3, logstd) spends 6 of 4 866 entries.  With the tower's block layout every parameter
# block (W1 / b1 / g1 / be1 / W2 / ... / W3 / b3 / logstd) has to point the reference's way on its own:
#   * a block whose reference update is non-trivial (rms|d_ref| of the block >= BLOCK_TRIVIAL x rms|d_ref| of the tower):
#       cosine(d_gpu, d_ref) over the block >= BLOCK_COS   (for a 1-entry block: same sign),
#   * a block the reference hardly moves: the engine must hardly move it either (rms|d_gpu - d_ref| below the same floor).
BLOCK_COS, BLOCK_TRIVIAL = 0.999, 0.05


def block_update_stats(theta0, got, ref, blocks):
    """{block: dict(cos, trivial, n, rms_ref, rms_err)} for `blocks` = {name: (offset, size)} of the flat vector."""
    theta0, got, ref = (np.asarray(a, dtype=np.float64).ravel() for a in (theta0, got, ref))
    dg, dr = got - theta0, ref - theta0
    rms_all = float(np.sqrt(np.mean(dr * dr)))
    out = {}
    for name, (off, n) in blocks.items():
        if n == 0:
            continue
        g, r = dg[off:off + n], dr[off:off + n]
        ng, nr = np.linalg.norm(g), np.linalg.norm(r)
        out[name] = dict(cos=float(g @ r / (ng * nr)) if ng > 0 and nr > 0 else (1.0 if ng == nr else 0.0), n=int(n),
                         rms_ref=float(nr / np.sqrt(n)), rms_err=float(np.linalg.norm(g - r) / np.sqrt(n)),
                         trivial=bool(nr / np.sqrt(n) < BLOCK_TRIVIAL * rms_all), floor=BLOCK_TRIVIAL * rms_all)
    return out


def assert_block_update_parity(theta0, got, ref, blocks, name="", cos_min=BLOCK_COS):
    bad = []
    st = block_update_stats(theta0, got, ref, blocks)
    for b, s in st.items():
        if s["trivial"]:
            if s["rms_err"] > s["floor"]:
                bad.append(f"{b} (n = {s['n']}): the reference hardly moves it (rms {s['rms_ref']:.2e}) but rms|d_gpu - d_ref| = "
                           f"{s['rms_err']:.2e} > {s['floor']:.2e}")
        elif s["cos"] < cos_min:
            bad.append(f"{b} (n = {s['n']}): cos(d_theta) = {s['cos']:.6f} < {cos_min}")
    assert not bad, f"{name}: per-block update parity FAILED: " + "; ".join(bad)
    return st


def assert_update_parity(theta0, got, ref, name="", cos_min=DTHETA_COS, frac_min=DTHETA_FRAC, blocks=None):
    """The parity bar on the weight update itself; raises AssertionError with the measured figures.  ``blocks`` ({name:
    (offset, size)}, e.g. ``tower_blocks(spec)``) adds the per-block bar above."""
    s = dtheta_stats(theta0, got, ref)
    assert s["med"] > 0, f"{name}: the reference update is empty - the golden case cannot pin anything"
    assert s["cos"] >= cos_min and s["frac"] >= frac_min, (
        f"{name}: update parity FAILED: cos(d_theta) = {s['cos']:.7f} (>= {cos_min}), entries within "
        f"{DTHETA_REL:g}|d_ref| + {DTHETA_REL:g} median|d_ref| = {s['frac']:.4f} (>= {frac_min}); "
        f"median|d_ref| = {s['med']:.3e}, worst |d_gpu - d_ref| = {s['worst']:.3e}, rel L2 = {s['rel_l2']:.3e}, n = {s['n']}")
    if blocks is not None:
        s["blocks"] = assert_block_update_parity(theta0, got, ref, blocks, name)
    return s


def assert_update_parity_rejects(theta0, got, ref, name="", blocks=None):
    """Negative control: `got` is a deliberately broken update; the bar above must refuse it."""
    try:
        assert_update_parity(theta0, got, ref, name, blocks=blocks)
    except AssertionError:
        return
    raise AssertionError(f"{name}: the update-parity bar ACCEPTED a deliberately broken update "
                         f"({dtheta_stats(theta0, got, ref)}) - it cannot fail, so it proves nothing")


def tower_blocks(spec):
    """(name, offset, size) of every block of a default tower's flat parameter vector (oracle TowerSpec.sizes order)."""
    out, o = {}, 0
    for name, sh in spec.sizes():
        n = int(np.prod(sh))
        out[name] = (o, n)
        o += n
    return out


def blocks_of(tower, recurrent=False):
    """Block layout of an engine tower (openrl_amd.modules.ppo_module.Tower) from its net descriptor - for golden cases that
    do not carry their buffers (the full-size ones regenerate them from a seed)."""
    n = tower.net
    if recurrent:
        from oracle import rnn_oracle as ro

        return tower_blocks(ro.RnnTowerSpec(int(n.obs_dim), int(n.n_out), int(n.head_kind)))
    return tower_blocks(po.TowerSpec(int(n.obs_dim), int(n.n_out), int(n.head_kind)))


def without_block_update(theta0, theta1, spec, block):
    """theta1 with `block` reset to theta0: the update a kernel would produce had it dropped that block's gradient."""
    off, n = tower_blocks(spec)[block]
    broken = np.array(theta1, copy=True)
    broken[off:off + n] = np.asarray(theta0)[off:off + n]
    return broken

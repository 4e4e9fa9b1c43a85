"""The power-of-two operand scaling of the towers' fp16 two-term split GEMMs (csrc/orl_mlp.h, ORL_TOWER_F16; the recurrent row
kernel's images: csrc/orl_rnn.h, ORL_RNN_L2_H2; round 6).

fp16 keeps 11 + 11 significand bits of an operand only inside [2^-14, 2^16): the update tower scales a tile of gradients by
its own maximum, the weight image by the image's maximum, and the wgrad accumulators follow a running scale.  These tests
drive the scaling logic over ranges the ordinary parity cases never visit:

* the critic's gradient is LINEAR in ``value_loss_coef``; a coefficient 2^-60 times smaller must give the bit-identical
  gradient times 2^-60 (every scale in the kernel is a power of two chosen from the data's exponent, so the fp16 terms are
  the same bits) - checked from 2^-80 to 2^40, 120 binades of gradient magnitude;
* W2 / b2 of both towers scaled by 2^-10 .. 2^+10 (the weight image's scale kw moves by as much), W2 = 0 (the image's
  maximum is zero) - one update against the oracle at the ordinary tolerances;
* rows whose gradients differ by many orders of magnitude inside one 16-row tile, and all-zero tiles.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests import test_ppo_update_gpu as TU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _critic_grad(g, coef):
    g2 = dict(g)
    g2["argv"] = str(g["argv"]) + " --value_loss_coef %r --use_max_grad_norm" % coef
    cfg, module, buf, algo = TU.build_engine(g2)
    assert not cfg.use_max_grad_norm and cfg.value_loss_coef == coef
    algo._advantages_and_records(buf)
    algo._info.zero_()
    M = buf.rewards.numel()
    algo._update_minibatch(buf, None, M, True)
    return module.models["critic"].grad.cpu().numpy().copy(), module.models["policy"].grad.cpu().numpy().copy()


@pytest.mark.parametrize("case", ["train_discrete", "train_gaussian"])
def test_critic_gradient_is_exactly_linear_in_a_power_of_two_loss_scale(case):
    g = H.load_golden(case)
    base_c, base_p = _critic_grad(g, 2.0 ** -20)
    assert np.isfinite(base_c).all() and np.abs(base_c).max() > 0
    for k in (-80, -50, 0, 40):
        got_c, got_p = _critic_grad(g, 2.0 ** k)
        want = np.ldexp(base_c.astype(np.float64), k + 20).astype(np.float32)
        assert np.array_equal(got_c, want), (k, float(np.abs(got_c - want).max()), float(np.abs(want).max()))
        assert np.array_equal(got_p, base_p), k  # the policy tower does not see the coefficient


def _scaled_w2(g, specs, k, zero=False):
    g2 = dict(g)
    for key, spec in (("theta_p0", specs[0]), ("theta_c0", specs[1])):
        th = torch.tensor(g[key]).clone()
        parts = spec.split(th)
        if zero:
            parts["W2"].zero_()
        else:
            parts["W2"].mul_(2.0 ** k)
            parts["b2"].mul_(2.0 ** k)
        g2[key] = th.numpy()
    return g2


@pytest.mark.parametrize("k", [-10, -4, 6, 10])
@pytest.mark.parametrize("case", ["train_discrete", "train_gaussian"])
def test_one_update_with_the_weight_image_scaled(case, k):
    """W2 and b2 times 2^k: z2 scales with them, LayerNorm 2 removes the scale again (up to its eps, which the oracle sees
    as well) - the fp16 image's own scale kw has to follow."""
    g = H.load_golden(case)
    TU.single_update_vs_oracle(_scaled_w2(g, H.case_specs(g), k))


@pytest.mark.parametrize("case", ["train_discrete"])
def test_one_update_with_a_zero_weight_image(case):
    g = H.load_golden(case)
    TU.single_update_vs_oracle(_scaled_w2(g, H.case_specs(g), 0, zero=True))


@pytest.mark.parametrize("case", ["train_discrete", "train_gaussian"])
def test_rows_of_very_different_gradient_magnitude_in_one_tile(case):
    """Returns (and with them the value error) of every third env blown up by 1e4 (the Huber loss' linear branch), every fourth
    env's rows inactive (zero gradient rows): tiles mix rows whose dz2 differ by orders of magnitude and hold all-zero rows."""
    g = dict(H.load_golden(case))
    ret = g["buf_returns"].copy()
    ret[:, ::3] *= 1e4
    g["buf_returns"] = ret
    act = g["buf_active_masks"].copy()
    act[:, ::4] = 0.0
    g["buf_active_masks"] = act
    g["argv"] = str(g["argv"]) + " --use_valuenorm false --use_adv_normalize false"
    TU.single_update_vs_oracle(g, info_rtol=5e-4)


# ---- the recurrent L = 2 row kernel over fp16 images (csrc/orl_rnn.h, ORL_RNN_L2_H2) -------------------------------------

def _scaled_gru(g, specs, k, names):
    g2 = dict(g)
    for key, spec in (("theta_p0", specs[0]), ("theta_c0", specs[1])):
        th = torch.tensor(g[key]).clone()
        parts = spec.split(th)
        for n in names:
            parts[n].mul_(2.0 ** k)
        g2[key] = th.numpy()
    return g2


@pytest.mark.parametrize("k", [-60, -30, 24])
def test_recurrent_update_with_a_scaled_loss(k):
    """data_chunk_length 2 = the register-resident row kernel: the gate deltas of a row are scaled by one power of two taken
    from the row's own maximum.  value_loss_coef x 2^k moves the critic's deltas by as many binades; one update against the
    oracle, gradients relative to their own largest entry."""
    from tests import test_rnn_kernels_gpu as TR

    g = dict(H.load_golden("train_recurrent"))
    g["argv"] = str(g["argv"]) + " --value_loss_coef %r" % (2.0 ** k)
    TR.rnn_update_vs_oracle(g)


@pytest.mark.parametrize("k,names", [(-8, ("Wih", "Whh", "bih", "bhh")), (7, ("Wih", "Whh", "bih", "bhh")), (-9, ("W2", "b2")),
                                     (8, ("W2", "b2"))])
def test_recurrent_update_with_scaled_weight_images(k, names):
    """The GRU's six matrices share ONE image scale (their products are added inside the gates), W2 has its own (LayerNorm 2
    takes fc2's scaled accumulators): both moved by 2^k."""
    from tests import rnn_helpers as RH
    from tests import test_rnn_kernels_gpu as TR

    g = H.load_golden("train_recurrent")
    TR.rnn_update_vs_oracle(_scaled_gru(g, RH.rnn_specs(g), k, names))

"""Recurrent (GRU) HIP kernels against the oracle / the golden vectors minted from the real reference
(SURVEY.md section 8a row a26): orl_rnn_act_step, orl_rnn_chunk_rows, orl_rnn_ppo_fwd_bwd, orl_rnn_ppo_apply."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import rnn_oracle as ro
from tests import helpers as H
from tests import rnn_helpers as RH

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")


def _nets(pspec, cspec):
    from openrl_amd import ops

    head = ops.HEAD_CATEGORICAL if pspec.head == po.HEAD_CATEGORICAL else ops.HEAD_GAUSSIAN
    return ops.net_desc(pspec.obs_dim, pspec.n_out, head), ops.net_desc(cspec.obs_dim, 1, ops.HEAD_VALUE)


def _t(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(DEV)


@pytest.mark.parametrize("case", RH.RNN_CASES)
def test_rnn_param_counts_and_probe_vs_reference(case):
    from openrl_amd import ops_rnn

    g = H.load_golden(case)
    pspec, cspec = RH.rnn_specs(g)
    pnet, cnet = _nets(pspec, cspec)
    assert ops_rnn.rnn_param_count(pnet) == g["theta_p1"].size == pspec.n_params()
    assert ops_rnn.rnn_param_count(cnet) == g["theta_c1"].size
    B = g["probe_policy_obs"].shape[0]
    a_w = g["probe_actions"].shape[1]
    values, actions, logp = torch.zeros(B, 1, device=DEV), torch.zeros(B, a_w, device=DEV), torch.zeros(B, a_w, device=DEV)
    hp1, hc1 = torch.zeros(B, 64, device=DEV), torch.zeros(B, 64, device=DEV)
    ops_rnn.rnn_act_step(pnet, _t(g["theta_p1"]), cnet, _t(g["theta_c1"]), _t(g["probe_policy_obs"]),
                         _t(g["probe_critic_obs"]), _t(g["probe_h"].reshape(B, 64)), _t(g["probe_hc"].reshape(B, 64)),
                         _t(g["probe_masks"].reshape(B)), None, B, True, 0, 0, 0, None, values, actions, logp, hp1, hc1)
    np.testing.assert_allclose(values.cpu().numpy(), g["probe_values"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(actions.cpu().numpy(), g["probe_actions"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g["probe_logp"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hp1.cpu().numpy(), g["probe_h1"].reshape(B, 64), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hc1.cpu().numpy(), g["probe_hc1"].reshape(B, 64), rtol=2e-4, atol=2e-5)
    # value-only call (PPOModule.get_values) and sampled call with forced uniforms vs the oracle's sampler
    v2, hc2 = torch.zeros(B, 1, device=DEV), torch.zeros(B, 64, device=DEV)
    ops_rnn.rnn_act_step(None, None, cnet, _t(g["theta_c1"]), None, _t(g["probe_critic_obs"]), None,
                         _t(g["probe_hc"].reshape(B, 64)), _t(g["probe_masks"].reshape(B)), None, B, True, 0, 0, 0, None,
                         v2, None, None, None, hc2)
    assert torch.equal(v2, values) and torch.equal(hc2, hc1)
    rs = np.random.RandomState(3)
    u = rs.rand(B, a_w).astype(np.float32) if pspec.head == po.HEAD_CATEGORICAL else rs.randn(B, a_w).astype(np.float32)
    ops_rnn.rnn_act_step(pnet, _t(g["theta_p1"]), cnet, _t(g["theta_c1"]), _t(g["probe_policy_obs"]),
                         _t(g["probe_critic_obs"]), _t(g["probe_h"].reshape(B, 64)), _t(g["probe_hc"].reshape(B, 64)),
                         _t(g["probe_masks"].reshape(B)), None, B, False, 0, 0, 0, _t(u), values, actions, logp, hp1, hc1)
    v_o, a_o, lp_o, _, _ = ro.get_actions(pspec, torch.tensor(g["theta_p1"]), cspec, torch.tensor(g["theta_c1"]),
                                          g["probe_policy_obs"], g["probe_critic_obs"], g["probe_h"], g["probe_hc"],
                                          g["probe_masks"], deterministic=False, forced_u=u)
    np.testing.assert_allclose(actions.cpu().numpy(), a_o, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_o, rtol=2e-4, atol=3e-5)


def test_chunk_rows_match_the_recurrent_generator_order():
    from openrl_amd import ops_rnn

    T, lanes, L = 7, 5, 2  # chunks straddle lanes (T odd)
    n_chunks = (T * lanes) // L
    chunks = torch.randperm(n_chunks)
    rows = torch.zeros(L * n_chunks, dtype=torch.int64, device=DEV)
    ops_rnn.rnn_chunk_rows(chunks.to(DEV), n_chunks, L, T, lanes, rows)
    # cast order row r = lane*T + t  ->  record row t*lanes + lane
    r = (chunks.numpy()[None, :] * L + np.arange(L)[:, None])
    want = (r % T) * lanes + (r // T)
    assert np.array_equal(rows.cpu().numpy().reshape(L, n_chunks), want)
    ops_rnn.rnn_chunk_rows(None, n_chunks, L, T, lanes, rows)
    r = (np.arange(n_chunks)[None, :] * L + np.arange(L)[:, None])
    assert np.array_equal(rows.cpu().numpy().reshape(L, n_chunks), (r % T) * lanes + (r // T))


def _records(buf, adv, K):
    fl = lambda x: x.reshape(-1, x.shape[-1])
    cols = [fl(buf["policy_obs"][:-1]), fl(buf["critic_obs"][:-1]), fl(buf["actions"]), fl(buf["action_log_probs"]),
            adv.reshape(-1, 1), fl(buf["value_preds"][:-1]), fl(buf["returns"][:-1]), fl(buf["active_masks"][:-1])]
    if K:
        cols.append(fl(buf["action_masks"][:-1]))
    rec = np.concatenate(cols, 1).astype(np.float32)
    pad = (-rec.shape[1]) % 4
    return np.pad(rec, ((0, 0), (0, pad)))


@pytest.mark.parametrize("case", RH.RNN_CASES)
def test_rnn_update_gradients_and_adam_step_vs_oracle(case):
    """First minibatch of the golden case: raw sums -> gradients, clip, Adam, train_info vs oracle autograd."""
    rnn_update_vs_oracle(H.load_golden(case))


def rnn_update_vs_oracle(g, grad_atol=3e-5):
    """(also driven by tests/test_layernorm_adversarial_gpu.py on golden cases with shifted initial weights)"""
    from openrl_amd import ops, ops_rnn

    cfg = H.case_cfg(g)
    hp_o = po.hyper_from_cfg(cfg)
    pspec, cspec = RH.rnn_specs(g)
    pnet, cnet = _nets(pspec, cspec)
    buf = H.case_buffer(g)
    T, N, A = buf["rewards"].shape[:3]
    lanes, L = N * A, cfg.data_chunk_length
    vn = po.ValueNormOracle()
    adv = po.advantages(buf["returns"], buf["value_preds"], buf["active_masks"], vn, hp_o.use_adv_normalize)
    rows_o = ro.buffer_rows(buf, adv)
    torch.manual_seed(int(g["perm_seed"]))
    chunks = ro.recurrent_chunk_order(T * lanes, L, cfg.num_mini_batch)[0]
    ptheta, ctheta = torch.tensor(g["theta_p0"]).clone(), torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    info_o, gp, gc, raw_p, raw_c = ro.ppo_update(hp_o, pspec, ptheta, cspec, ctheta, padam, cadam, vn,
                                                  ro.chunk_sample(rows_o, chunks, L))
    # ---- engine
    K = pspec.n_out if pspec.head == po.HEAD_CATEGORICAL else 0
    rec = _t(_records(buf, adv, K))
    assert rec.shape[1] == ops.record_width(pspec.obs_dim, cspec.obs_dim, buf["actions"].shape[-1], K)
    n_chunks = len(chunks)
    rows = torch.zeros(L * n_chunks, dtype=torch.int64, device=DEV)
    ops_rnn.rnn_chunk_rows(_t(chunks, torch.int64), n_chunks, L, T, lanes, rows)
    vn_state = torch.zeros(3, device=DEV)
    mom, scratch = torch.zeros(3, dtype=torch.float64, device=DEV), torch.zeros(512, dtype=torch.float64, device=DEV)
    ret_col = pspec.obs_dim + cspec.obs_dim + 2 * buf["actions"].shape[-1] + 2
    ops.minibatch_moments(rec, ret_col, rows, L * n_chunks, scratch, mom)
    ops.valuenorm_update(vn_state, mom, 0.99999)
    np.testing.assert_allclose(vn_state.cpu().numpy(), vn.state(), rtol=1e-6)
    hp = ops.make_hparams(cfg)
    th_p, th_c = _t(g["theta_p0"]), _t(g["theta_c0"])
    ws = torch.zeros(ops_rnn.rnn_workspace_floats(pnet, cnet, n_chunks, L), device=DEV)
    raw_np, raw_nc = ops_rnn.rnn_raw_grad_count(pnet), ops_rnn.rnn_raw_grad_count(cnet)
    sums = torch.zeros(raw_np + raw_nc + 2 * ops.N_STATS, device=DEV)
    masks = _t(buf["masks"].reshape(T + 1, lanes))
    h_p, h_c = _t(buf["rnn_states"].reshape(T + 1, lanes, 64)), _t(buf["rnn_states_critic"].reshape(T + 1, lanes, 64))
    ops_rnn.rnn_ppo_fwd_bwd(pnet, th_p, cnet, th_c, rec, rows, masks, h_p, h_c, n_chunks, L, vn_state, hp, ws, sums)
    st_p = sums[raw_np:raw_np + ops.N_STATS].cpu().numpy()
    assert st_p[1] == L * n_chunks  # rows counted
    from openrl_amd._native import AdamState, fptr

    gr_p, gr_c = torch.zeros_like(th_p), torch.zeros_like(th_c)
    m_p, v_p, m_c, v_c = (torch.zeros_like(th_p), torch.zeros_like(th_p), torch.zeros_like(th_c), torch.zeros_like(th_c))
    info = torch.zeros(8, device=DEV)
    ops_rnn.rnn_ppo_apply(pnet, cnet, sums, hp,
                          AdamState(fptr(th_p), fptr(gr_p), fptr(m_p), fptr(v_p), cfg.lr, cfg.opti_eps, cfg.weight_decay, 1),
                          AdamState(fptr(th_c), fptr(gr_c), fptr(m_c), fptr(v_c), cfg.critic_lr, cfg.opti_eps,
                                    cfg.weight_decay, 1), info, torch.zeros(512, device=DEV))
    np.testing.assert_allclose(gr_p.cpu().numpy(), gp, rtol=3e-3, atol=grad_atol * np.abs(gp).max() + 1e-7)
    np.testing.assert_allclose(gr_c.cpu().numpy(), gc, rtol=3e-3, atol=grad_atol * np.abs(gc).max() + 1e-7)
    np.testing.assert_allclose(info[:6].cpu().numpy(), np.array([info_o[k] for k in KEYS]), rtol=3e-4, atol=3e-5)
    # Adam: the oracle's step applied to the ENGINE's clipped gradients must land on the engine's parameters
    for th_e, gr_e, th0, lr in ((th_p, gr_p, g["theta_p0"], cfg.lr), (th_c, gr_c, g["theta_c0"], cfg.critic_lr)):
        th = torch.tensor(th0).clone()
        po.AdamOracle(th.numel(), lr, cfg.opti_eps, cfg.weight_decay).step(th, gr_e.cpu())
        np.testing.assert_allclose(th_e.cpu().numpy(), th.numpy(), rtol=1e-5, atol=1e-7)


def test_rnn_act_step_large_batch_path_equals_small_batch_path():
    """>= 128 tiles with both towers runs the LDS-staged workgroup kernel; the same rows in chunks of 64 tiles run
    the per-tile kernel that reads weights through L2.  Same arithmetic order -> identical results."""
    from openrl_amd import ops, ops_rnn

    B, Dp, Dc, K = 4099, 18, 54, 5  # ragged: last tile partly filled
    pnet, cnet = ops.net_desc(Dp, K, ops.HEAD_CATEGORICAL), ops.net_desc(Dc, 1, ops.HEAD_VALUE)
    g = torch.Generator(device=DEV).manual_seed(3)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    thp, thc = 0.2 * r(ops_rnn.rnn_param_count(pnet)), 0.2 * r(ops_rnn.rnn_param_count(cnet))
    po_, co_, hp, hc = r(B, Dp), r(B, Dc), 0.5 * r(B, 64), 0.5 * r(B, 64)
    mk = (torch.rand(B, device=DEV, generator=g) > 0.2).float()
    u = torch.rand(B, 1, device=DEV, generator=g)
    out = lambda: (torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV),
                   torch.zeros(B, 64, device=DEV), torch.zeros(B, 64, device=DEV))
    v1, a1, l1, hp1, hc1 = out()
    ops_rnn.rnn_act_step(pnet, thp, cnet, thc, po_, co_, hp, hc, mk, None, B, False, 0, 0, 0, u, v1, a1, l1, hp1, hc1)
    v2, a2, l2, hp2, hc2 = out()
    for lo in range(0, B, 1024):
        hi = min(lo + 1024, B)
        ops_rnn.rnn_act_step(pnet, thp, cnet, thc, po_[lo:hi], co_[lo:hi], hp[lo:hi], hc[lo:hi], mk[lo:hi], None, hi - lo,
                             False, 0, 0, 0, u[lo:hi], v2[lo:hi], a2[lo:hi], l2[lo:hi], hp2[lo:hi], hc2[lo:hi])
    for x, y in ((v1, v2), (l1, l2), (hp1, hp2), (hc1, hc2)):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    assert (a1 == a2).float().mean() > 0.999 and a1.max() <= K - 1


@pytest.mark.parametrize("head,n_out,Dp,Dc,T,lanes", [
    ("cat", 5, 18, 54, 25, 96),     # cfg4's towers: wide categorical head (MFMA head + distributed loss), Dict obs
    ("cat", 2, 4, 4, 9, 37),        # narrow categorical head with action masks, ragged tile (166 chunks)
    ("box", 3, 11, 11, 7, 50),      # Gaussian head (per-dimension record columns are read through the fallback)
])
def test_l2_row_kernel_equals_the_recompute_kernel(head, n_out, Dp, Dc, T, lanes):
    """Round 5: with data_chunk_length == 2 orl_rnn_ppo_fwd_bwd runs the register-resident row kernel (csrc/orl_rnn_l2.h,
    hparams.reserved == 0); reserved & 4 forces the forward-sweep + recompute kernel of rounds 1 - 4.  Same inputs (random
    buffers, stored states, 10 % zero masks, a shuffled chunk order, a chunk count that does not fill the last tile) -> the
    same raw sums: every weight-gradient sum, bias sum, dlogstd and the statistics row, up to fp32 summation order."""
    from openrl_amd import ops, ops_rnn
    from openrl_amd.configs.config import default_cfg

    L = 2
    rs = np.random.RandomState(11)
    a_w = 1 if head == "cat" else n_out
    K = n_out if head == "cat" else 0
    pnet = ops.net_desc(Dp, n_out, ops.HEAD_CATEGORICAL if head == "cat" else ops.HEAD_GAUSSIAN)
    cnet = ops.net_desc(Dc, 1, ops.HEAD_VALUE)
    M = T * lanes
    R = ops.record_width(Dp, Dc, a_w, K)
    rec = np.zeros((M, R), np.float32)
    o = 0
    rec[:, o:o + Dp] = rs.randn(M, Dp); o += Dp
    rec[:, o:o + Dc] = rs.randn(M, Dc); o += Dc
    rec[:, o:o + a_w] = rs.randint(0, n_out, (M, 1)) if head == "cat" else rs.randn(M, a_w); o += a_w
    rec[:, o:o + a_w] = np.log(1.0 / max(n_out, 2)) + 0.1 * rs.randn(M, a_w); o += a_w
    rec[:, o] = rs.randn(M); o += 1                       # advantage
    rec[:, o] = 0.3 * rs.randn(M); o += 1                 # value prediction
    rec[:, o] = 0.5 * rs.randn(M); o += 1                 # return
    rec[:, o] = (rs.rand(M) > 0.05).astype(np.float32); o += 1   # active mask
    if K:
        am = (rs.rand(M, K) > 0.3).astype(np.float32)
        am[np.arange(M), rec[:, Dp + Dc].astype(int)] = 1.0    # the taken action is legal
        rec[:, o:o + K] = am
    masks = (rs.rand(T + 1, lanes) > 0.1).astype(np.float32)
    h_p, h_c = 0.4 * rs.randn(T + 1, lanes, 64).astype(np.float32), 0.4 * rs.randn(T + 1, lanes, 64).astype(np.float32)
    n_chunks = M // L
    chunks = rs.permutation(n_chunks)
    rows = torch.zeros(L * n_chunks, dtype=torch.int64, device=DEV)
    ops_rnn.rnn_chunk_rows(_t(chunks, torch.int64), n_chunks, L, T, lanes, rows)
    thp = _t(0.15 * rs.randn(ops_rnn.rnn_param_count(pnet)))
    thc = _t(0.15 * rs.randn(ops_rnn.rnn_param_count(cnet)))
    vn_state = _t(np.array([0.1, 0.5, 1.0], np.float32))
    cfg = default_cfg(["--use_recurrent_policy", "true", "--data_chunk_length", "2"])
    raw_np, raw_nc = ops_rnn.rnn_raw_grad_count(pnet), ops_rnn.rnn_raw_grad_count(cnet)
    out = {}
    for mode in ("fp32", "fp32_recompute"):
        cfg.amd_rnn_gemm = mode
        hp = ops.make_hparams(cfg, recurrent=True)
        assert hp.reserved == (0 if mode == "fp32" else 4)
        ws = torch.zeros(ops_rnn.rnn_workspace_floats(pnet, cnet, n_chunks, L), device=DEV)
        sums = torch.zeros(raw_np + raw_nc + 2 * ops.N_STATS, device=DEV)
        ops_rnn.rnn_ppo_fwd_bwd(pnet, thp, cnet, thc, _t(rec), rows, _t(masks), _t(h_p), _t(h_c), n_chunks, L, vn_state, hp,
                                ws, sums)
        out[mode] = sums.cpu().numpy()
    a, b = out["fp32"], out["fp32_recompute"]
    assert np.isfinite(a).all() and np.abs(b).max() > 0
    for lo, hi, name in ((0, raw_np, "policy raw sums"), (raw_np, raw_np + ops.N_STATS, "policy statistics"),
                         (raw_np + ops.N_STATS, raw_np + ops.N_STATS + raw_nc, "critic raw sums"),
                         (raw_np + ops.N_STATS + raw_nc, a.size, "critic statistics")):
        np.testing.assert_allclose(a[lo:hi], b[lo:hi], rtol=2e-4, atol=2e-5 * np.abs(b[lo:hi]).max() + 1e-7, err_msg=name)

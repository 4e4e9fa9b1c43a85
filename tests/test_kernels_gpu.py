"""Kernel-level parity: each C-ABI entry point against the oracle / golden vectors.  Needs a MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import philox as px
from oracle import ppo_oracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(DEV).contiguous()


@pytest.fixture(scope="module")
def ops():
    from openrl_amd import ops as _ops

    return _ops


# --------------------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("use_gae", [True, False])
@pytest.mark.parametrize("proper", [False, True])
@pytest.mark.parametrize("use_vn", [False, True])
def test_gae_scan_bit_exact_vs_reference(ops, use_gae, proper, use_vn):
    g = H.load_golden("gae")
    T, N, A, _ = g["rand_rewards"].shape
    rewards, vp, masks, bad = dev(g["rand_rewards"]), dev(g["rand_value_preds"]), dev(g["rand_masks"]), dev(
        g["rand_bad_masks"])
    nv = dev(g["rand_next_value"])
    ret = torch.zeros_like(vp)
    vn = dev(g["rand_vn_state"]) if use_vn else None
    ops.gae_scan(rewards, vp, masks, bad, nv, vn, ret, 0.99, 0.95, use_gae, proper)
    tag = "rand_g%d_p%d_v%d" % (use_gae, proper, use_vn)
    got = ret.cpu().numpy()
    # tolerance stated: bit-exact when ValueNorm is off (pure fp32 mul/add in the reference's order);
    # with ValueNorm the de-normalisation sqrt/div may differ by 1 ulp -> 1e-6 relative
    if use_vn:
        np.testing.assert_allclose(got, g[tag + "_returns"], rtol=1e-6, atol=1e-6)
    else:
        assert np.array_equal(got, g[tag + "_returns"])
    assert np.array_equal(vp.cpu().numpy(), g[tag + "_value_preds"])


def test_gae_known_answer_vector(ops):
    g = H.load_golden("gae")
    rewards = dev(np.array([[1, .5], [1, -1], [1, 2], [1, .25]], np.float32).reshape(4, 2, 1, 1))
    vp = torch.zeros(5, 2, 1, 1, device=DEV)
    vp[:4] = dev(np.array([[.5, .1], [.4, -.2], [.3, .7], [.2, 0]], np.float32).reshape(4, 2, 1, 1))
    masks = dev(np.array([[1, 1], [1, 1], [1, 0], [1, 1], [1, 1]], np.float32).reshape(5, 2, 1, 1))
    bad = dev(np.array([[1, 1], [1, 1], [1, 1], [1, 0], [1, 1]], np.float32).reshape(5, 2, 1, 1))
    nv = dev(np.array([.1, .9], np.float32).reshape(2, 1, 1))
    for proper in (False, True):
        ret = torch.zeros_like(vp)
        ops.gae_scan(rewards, vp.clone(), masks, bad, nv, None, ret, 0.99, 0.95, True, proper)
        assert np.array_equal(ret.cpu().numpy(), g["kat_proper%d_returns" % proper])


# T > 128 crosses the kernel's LDS time-chunk boundary (configs 3 and 5 roll out 200 steps); L not a multiple of 16
@pytest.mark.parametrize("T,L", [(1, 1), (37, 70), (128, 4096), (200, 1024), (129, 17), (300, 33)])
def test_gae_ragged_and_full_size(ops, T, L):
    rs = np.random.RandomState(T * 1000 + L)
    rewards = rs.randn(T, L, 1, 1).astype(np.float32)
    vp = rs.randn(T + 1, L, 1, 1).astype(np.float32)
    masks = (rs.rand(T + 1, L, 1, 1) > 0.1).astype(np.float32)
    nv = rs.randn(L, 1, 1).astype(np.float32)
    want, want_vp = po.compute_returns(rewards, vp, masks, None, nv, 0.99, 0.95, True, False, None)
    d_vp, ret = dev(vp), torch.zeros(T + 1, L, 1, 1, device=DEV)
    ops.gae_scan(dev(rewards), d_vp, dev(masks), None, dev(nv), None, ret, 0.99, 0.95, True, False)
    assert np.array_equal(ret.cpu().numpy(), want)
    assert np.array_equal(d_vp.cpu().numpy(), want_vp)


@pytest.mark.parametrize("use_gae", [True, False])
@pytest.mark.parametrize("proper", [False, True])
@pytest.mark.parametrize("use_vn", [False, True])
def test_gae_long_rollout_all_variants(ops, use_gae, proper, use_vn):
    """All four compute_returns variants across the 128-step LDS chunk boundary (T = 200, ragged lane count), against
    the restatement that the golden vectors pin (tests/test_oracle_cpu.py)."""
    T, L = 200, 50
    rs = np.random.RandomState(77 + 2 * use_gae + proper)
    rewards = rs.randn(T, L, 1, 1).astype(np.float32)
    vp = rs.randn(T + 1, L, 1, 1).astype(np.float32)
    masks = (rs.rand(T + 1, L, 1, 1) > 0.05).astype(np.float32)
    bad = (rs.rand(T + 1, L, 1, 1) > 0.05).astype(np.float32)
    nv = rs.randn(L, 1, 1).astype(np.float32)
    vn = None
    if use_vn:
        vn = po.ValueNormOracle()
        vn.set_state([0.3e-3, 2.5e-3, 1.2e-3])
    want, want_vp = po.compute_returns(rewards, vp, masks, bad if proper else None, nv, 0.99, 0.95, use_gae, proper, vn)
    d_vp, ret = dev(vp), torch.zeros(T + 1, L, 1, 1, device=DEV)
    ops.gae_scan(dev(rewards), d_vp, dev(masks), dev(bad) if proper else None, dev(nv),
                 dev(vn.state()) if use_vn else None, ret, 0.99, 0.95, use_gae, proper)
    assert np.array_equal(ret.cpu().numpy(), want)
    assert np.array_equal(d_vp.cpu().numpy(), want_vp)


def test_grouped_act_step_equals_one_launch_per_policy(ops):
    """orl_act_step_grouped (a pool of policies in one launch) == orl_act_step per row group, bit for bit."""
    rs = np.random.RandomState(3)
    D, K, B, G = 18, 9, 100, 32          # 4 groups: 32 + 32 + 32 + 4 rows
    pnet = ops.net_desc(D, K, ops.HEAD_CATEGORICAL)
    thetas = dev((0.3 * rs.randn(4, ops.param_count(pnet))).astype(np.float32))
    obs = dev(rs.randn(B, D).astype(np.float32))
    masks = (rs.rand(B, K) > 0.4).astype(np.float32)
    masks[:, 0] = 1.0
    masks = dev(masks)
    a1, l1 = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    a2, l2 = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    ops.act_step_grouped(pnet, thetas, G, obs, masks, B, False, 77, 0, 5, a1, l1)
    for g in range(4):
        r0, r1 = g * G, min((g + 1) * G, B)
        ops.act_step(pnet, thetas[g], None, None, obs[r0:r1], None, masks[r0:r1], r1 - r0, False, 77, r0, 5, None, None,
                     a2[r0:r1], l2[r0:r1])
    assert torch.equal(a1, a2) and torch.equal(l1, l2)
    assert len(torch.unique(a1)) > 3


@pytest.mark.parametrize("D,B", [(4, 1000), (18, 4099), (54, 33)])
def test_batched_critic_values_equal_the_act_step_critic(ops, D, B):
    """orl_critic_values (the fused rollout's value pass over all T+1 slots) == orl_act_step's critic wave, bit for bit."""
    rs = np.random.RandomState(D)
    cnet = ops.net_desc(D, 1, ops.HEAD_VALUE)
    n_par = D * 64 + 64 * 3 + 64 * 64 + 64 * 3 + 64 + 1
    theta = dev((0.2 * rs.randn(n_par)).astype(np.float32))
    obs = dev(rs.randn(B, D).astype(np.float32))
    want, got = torch.empty(B, 1, device=DEV), torch.full((B,), np.nan, device=DEV)
    pnet = ops.net_desc(D, 2, ops.HEAD_CATEGORICAL)  # descriptor only: no policy parameters are passed
    ops.act_step(pnet, None, cnet, theta, None, obs, None, B, True, 0, 0, 0, None, want, None, None)
    ops.critic_values(cnet, theta, obs, got)
    assert torch.equal(got, want.view(-1))


# --------------------------------------------------------------------------------------------- K7
@pytest.mark.parametrize("T", [11, 150])
@pytest.mark.parametrize("use_adv_norm", [False, True])
@pytest.mark.parametrize("use_vn", [False, True])
def test_advantage_normalisation_and_record_packing(ops, use_adv_norm, use_vn, T):
    rs = np.random.RandomState(5)
    N, A, Dp, Dc, K = 13, 2, 3, 5, 4
    L = N * A
    rewards = rs.randn(T, N, A, 1).astype(np.float32)
    vp = rs.randn(T + 1, N, A, 1).astype(np.float32)
    masks = (rs.rand(T + 1, N, A, 1) > 0.1).astype(np.float32)
    active = (rs.rand(T + 1, N, A, 1) > 0.2).astype(np.float32)
    nv = rs.randn(N, A, 1).astype(np.float32)
    vn = None
    if use_vn:
        vn = po.ValueNormOracle()
        vn.set_state([0.3e-3, 2.5e-3, 1.2e-3])
    want_ret, want_vp = po.compute_returns(rewards, vp, masks, None, nv, 0.99, 0.95, True, False, vn)
    want_adv = po.advantages(want_ret, want_vp, active, vn, use_adv_norm)

    pobs = rs.randn(T + 1, N, A, Dp).astype(np.float32)
    cobs = rs.randn(T + 1, N, A, Dc).astype(np.float32)
    act = rs.randint(0, K, (T, N, A, 1)).astype(np.float32)
    lp = rs.randn(T, N, A, 1).astype(np.float32)
    am = (rs.rand(T + 1, N, A, K) > 0.3).astype(np.float32)

    d_vp, d_ret, d_active = dev(vp), torch.zeros(T + 1, N, A, 1, device=DEV), dev(active)
    adv = torch.empty(T, N, A, 1, device=DEV)
    partials = torch.zeros(ops.gae_max_partials(T, L), 8, dtype=torch.float64, device=DEV)
    vn_d = dev(vn.state()) if use_vn else None
    n_part = ops.gae_scan(dev(rewards), d_vp, dev(masks), None, dev(nv), vn_d, d_ret, 0.99, 0.95, True, False,
                          active_masks=d_active, adv_raw=adv, stat_partials=partials)
    R = ops.record_width(Dp, Dc, 1, K)
    rec = torch.empty(T * L, R, device=DEV)
    d_pobs, d_cobs, d_act, d_lp, d_am = dev(pobs), dev(cobs), dev(act), dev(lp), dev(am)
    from openrl_amd._native import PackSrc, fptr

    src = PackSrc(fptr(d_pobs), fptr(d_cobs), fptr(d_act), fptr(d_lp), fptr(d_vp), fptr(d_ret), fptr(d_active),
                  fptr(d_am), Dp, Dc, 1, K)
    stats = torch.zeros(11, dtype=torch.float64, device=DEV)
    ops.adv_normalize_pack(adv, partials, n_part, T, L, use_adv_norm, stats, src, rec)
    got_adv = adv.cpu().numpy()
    # fp32 tolerance: reductions differ in order (double tree on device vs numpy pairwise fp32)
    np.testing.assert_allclose(got_adv, want_adv, rtol=2e-5, atol=2e-5)
    r = rec.cpu().numpy()
    M = T * L
    assert np.array_equal(r[:, :Dp], pobs[:-1].reshape(M, Dp))
    assert np.array_equal(r[:, Dp:Dp + Dc], cobs[:-1].reshape(M, Dc))
    o = Dp + Dc
    assert np.array_equal(r[:, o], act.reshape(M))
    assert np.array_equal(r[:, o + 1], lp.reshape(M))
    assert np.array_equal(r[:, o + 2], got_adv.reshape(M))
    assert np.array_equal(r[:, o + 3], d_vp.cpu().numpy()[:-1].reshape(M))
    assert np.array_equal(r[:, o + 4], d_ret.cpu().numpy()[:-1].reshape(M))
    assert np.array_equal(r[:, o + 5], active[:-1].reshape(M))
    assert np.array_equal(r[:, o + 6:o + 6 + K], am[:-1].reshape(M, K))
    st = stats.cpu().numpy()
    assert st[2] == M and st[5] == active[:-1].sum()


# --------------------------------------------------------------------------------------------- K5 / K8
def test_buffer_insert_masks_bit_exact(ops):
    from openrl_amd._native import BufferPtrs, fptr

    rs = np.random.RandomState(3)
    T, N, A, Dp, Dc, K = 4, 9, 3, 5, 7, 6
    z = lambda *s: torch.zeros(*s, device=DEV)
    pobs, cobs = z(T + 1, N, A, Dp), z(T + 1, N, A, Dc)
    rew = z(T, N, A, 1)
    masks, bad, active = torch.ones(T + 1, N, A, 1, device=DEV), torch.ones(T + 1, N, A, 1, device=DEV), torch.ones(
        T + 1, N, A, 1, device=DEV)
    amask = torch.ones(T + 1, N, A, K, device=DEV)
    bp = BufferPtrs(fptr(pobs), fptr(cobs), fptr(rew), fptr(masks), fptr(bad), fptr(active), fptr(amask), T, N, A, Dp,
                    Dc, K)
    step = 2
    dones = rs.rand(N, A) < 0.5
    dones[0] = True
    dones[1] = False
    badt = rs.rand(N, A) < 0.3
    nob_p, nob_c = rs.randn(N, A, Dp).astype(np.float32), rs.randn(N, A, Dc).astype(np.float32)
    r = rs.randn(N, A, 1).astype(np.float32)
    nam = (rs.rand(N, A, K) > 0.5).astype(np.float32)
    ops.buffer_insert(bp, step, dev(nob_p), dev(nob_c), dev(r), dev(dones, torch.uint8), dev(badt, torch.uint8),
                      dev(nam))
    # reference semantics (onpolicy_driver.py:91-138)
    dones_env = np.all(dones, axis=1)
    w_masks = np.ones((N, A, 1), np.float32); w_masks[dones_env] = 0
    w_active = np.ones((N, A, 1), np.float32); w_active[dones] = 0; w_active[dones_env] = 1
    w_bad = np.where(badt, 0.0, 1.0).astype(np.float32)[..., None]
    assert np.array_equal(masks[step + 1].cpu().numpy(), w_masks)
    assert np.array_equal(active[step + 1].cpu().numpy(), w_active)
    assert np.array_equal(bad[step + 1].cpu().numpy(), w_bad)
    assert np.array_equal(pobs[step + 1].cpu().numpy(), nob_p)
    assert np.array_equal(cobs[step + 1].cpu().numpy(), nob_c)
    assert np.array_equal(rew[step].cpu().numpy(), r)
    assert np.array_equal(amask[step + 1].cpu().numpy(), nam)
    assert float(masks[step].min()) == 1.0 and float(pobs[step].abs().max()) == 0.0


def test_gather_minibatch_bit_exact(ops):
    rs = np.random.RandomState(1)
    M = 1000
    arrays = [rs.randn(M, w).astype(np.float32) for w in (4, 4, 64, 1, 1, 1, 1, 2, 1, 1, 5, 3)]
    torch.manual_seed(11)
    idx = torch.randperm(M)[:333]
    outs = ops.gather_minibatch([dev(a) for a in arrays], idx.to(DEV))
    for a, o in zip(arrays, outs):
        assert np.array_equal(o.cpu().numpy(), a[idx.numpy()])
    assert ops.gather_minibatch([dev(arrays[0])], torch.empty(0, dtype=torch.int64, device=DEV))[0].shape == (0, 4)


@pytest.mark.parametrize("n", [1, 2, 1000, 524288])
def test_perm_feistel_matches_oracle_and_is_bijective(ops, n):
    got = ops.perm_feistel(n, 77, 3, DEV).cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(n))
    if n <= 1000:
        assert np.array_equal(got, px.feistel_perm(n, 77, 3))


def test_perm_feistel_vn_equals_the_two_separate_launches(ops):
    """orl_perm_feistel_vn = orl_perm_feistel + orl_valuenorm_update in one launch (bit-exact both ways)."""
    n = 70_001
    mom = torch.tensor([1234.5, 98765.25, 4096.0], dtype=torch.float64, device=DEV)
    a = torch.tensor([0.3, 1.7, 0.25], device=DEV)
    b = a.clone()
    for k in range(3):
        pa = ops.perm_feistel(n, 5, 10 + k, DEV, vn=(a, mom, 0.99999))
        pb = ops.perm_feistel(n, 5, 10 + k, DEV)
        ops.valuenorm_update(b, mom, 0.99999)
        assert torch.equal(pa, pb) and torch.equal(a, b)
    assert not torch.equal(a, torch.tensor([0.3, 1.7, 0.25], device=DEV))


# --------------------------------------------------------------------------------------------- K1-K4
@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_act_step_deterministic_probe_vs_reference(ops, case):
    g = H.load_golden(case)
    pspec, cspec = H.case_specs(g)
    pnet = ops.net_desc(pspec.obs_dim, pspec.n_out, pspec.head)
    cnet = ops.net_desc(cspec.obs_dim, 1, ops.HEAD_VALUE)
    obs = dev(g["probe_obs"])
    B = obs.shape[0]
    a_w = g["probe_actions"].shape[1]
    values, actions, logp = torch.empty(B, 1, device=DEV), torch.empty(B, a_w, device=DEV), torch.empty(B, a_w,
                                                                                                      device=DEV)
    am = dev(g["probe_masks"]) if "probe_masks" in g else None
    ops.act_step(pnet, dev(g["theta_p1"]), cnet, dev(g["theta_c1"]), obs, obs, am, B, True, 0, 0, 0, None, values,
                 actions, logp)
    # fp32 tolerance: MFMA k-order differs from the CPU GEMM's
    np.testing.assert_allclose(values.cpu().numpy(), g["probe_values"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(actions.cpu().numpy(), g["probe_actions"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g["probe_logp"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", ["train_discrete", "train_discrete_masks", "train_gaussian"])
@pytest.mark.parametrize("B", [1, 16, 1000])
def test_act_step_teacher_forced_sampling_vs_oracle(ops, case, B):
    g = H.load_golden(case)
    pspec, cspec = H.case_specs(g)
    rs = np.random.RandomState(B)
    obs = rs.randn(B, pspec.obs_dim).astype(np.float32)
    a_w = 1 if pspec.head == po.HEAD_CATEGORICAL else pspec.n_out
    if pspec.head == po.HEAD_CATEGORICAL:
        forced = rs.rand(B, 1).astype(np.float32)
        am = (rs.rand(B, pspec.n_out) > 0.3).astype(np.float32) if "probe_masks" in g else None
        if am is not None:
            am[:, 0] = 1
    else:
        forced = rs.randn(B, a_w).astype(np.float32)
        am = None
    want_v, want_a, want_lp = po.get_actions(pspec, torch.tensor(g["theta_p1"]), cspec, torch.tensor(g["theta_c1"]),
                                             obs, obs, am, False, forced)
    pnet = ops.net_desc(pspec.obs_dim, pspec.n_out, pspec.head)
    cnet = ops.net_desc(cspec.obs_dim, 1, ops.HEAD_VALUE)
    values, actions, logp = torch.empty(B, 1, device=DEV), torch.empty(B, a_w, device=DEV), torch.empty(B, a_w,
                                                                                                      device=DEV)
    ops.act_step(pnet, dev(g["theta_p1"]), cnet, dev(g["theta_c1"]), dev(obs), dev(obs),
                 None if am is None else dev(am), B, False, 0, 0, 0, dev(forced), values, actions, logp)
    np.testing.assert_allclose(values.cpu().numpy(), want_v, rtol=1e-4, atol=1e-5)
    got_a = actions.cpu().numpy()
    if pspec.head == po.HEAD_CATEGORICAL:
        same = got_a[:, 0] == want_a[:, 0]
        assert same.mean() >= 0.995  # a uniform within 1e-6 of a CDF edge may flip
        np.testing.assert_allclose(logp.cpu().numpy()[same], want_lp[same], rtol=1e-4, atol=1e-5)
        if am is not None:
            assert np.all(am[np.arange(B), got_a[:, 0].astype(int)] == 1), "sampled a masked action"
    else:
        np.testing.assert_allclose(got_a, want_a, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(logp.cpu().numpy(), want_lp, rtol=1e-4, atol=2e-5)


def test_act_step_philox_stream_matches_oracle(ops):
    """Untethered sampling: device Philox uniforms == oracle Philox uniforms, so actions are reproducible."""
    g = H.load_golden("train_discrete")
    pspec, cspec = H.case_specs(g)
    B, seed, row0, step = 300, 1234567, 17, 5
    rs = np.random.RandomState(0)
    obs = rs.randn(B, 4).astype(np.float32)
    x, _, _, _ = px.philox4x32_10(seed, (np.arange(B) + row0).astype(np.uint32), 0, step, 0)
    u = px.u01(x).reshape(B, 1)
    _, want_a, _ = po.get_actions(pspec, torch.tensor(g["theta_p1"]), cspec, torch.tensor(g["theta_c1"]), obs, obs,
                                  None, False, u)
    pnet, cnet = ops.net_desc(4, 2, ops.HEAD_CATEGORICAL), ops.net_desc(4, 1, ops.HEAD_VALUE)
    values, actions, logp = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    ops.act_step(pnet, dev(g["theta_p1"]), cnet, dev(g["theta_c1"]), dev(obs), dev(obs), None, B, False, seed, row0,
                 step, None, values, actions, logp)
    assert (actions.cpu().numpy()[:, 0] == want_a[:, 0]).mean() >= 0.995


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_evaluate_actions_vs_oracle(ops, case):
    """PPOModule.evaluate_actions (ppo_module.py:149-193): log-prob of stored actions, masked-mean entropy, values."""
    g = H.load_golden(case)
    cfg = H.case_cfg(g)
    pspec, cspec = H.case_specs(g)
    b = H.case_buffer(g)
    fr = po.flat_rows
    obs, act, active = fr(b["policy_obs"][:-1]), fr(b["actions"]), fr(b["active_masks"][:-1])
    am = None if b["action_masks"] is None else fr(b["action_masks"][:-1])
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    with torch.no_grad():
        want_lp, want_ent = po.evaluate_actions(pspec, torch.tensor(g["theta_p1"]), t(obs), t(act), t(am), t(active),
                                                cfg.use_policy_active_masks)
        want_v = po.tower_forward(cspec, torch.tensor(g["theta_c1"]), t(obs))
    B, a_w = obs.shape[0], act.shape[1]
    pnet, cnet = ops.net_desc(pspec.obs_dim, pspec.n_out, pspec.head), ops.net_desc(cspec.obs_dim, 1, ops.HEAD_VALUE)
    values, logp = torch.empty(B, 1, device=DEV), torch.empty(B, a_w, device=DEV)
    ent_rows, ent = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    ops.evaluate_actions(pnet, dev(g["theta_p1"]), cnet, dev(g["theta_c1"]), dev(obs), dev(obs), dev(act),
                         None if am is None else dev(am), dev(active) if cfg.use_policy_active_masks else None, B,
                         values, logp, ent_rows, ent)
    np.testing.assert_allclose(logp.cpu().numpy(), want_lp.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(values.cpu().numpy(), want_v.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ent.item(), want_ent.item(), rtol=1e-5, atol=1e-6)

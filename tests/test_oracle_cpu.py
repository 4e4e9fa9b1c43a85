"""Pin the oracle (oracle/*.py) against the golden vectors minted from the REAL reference
(oracle/gen_golden.py) and against published known-answer vectors.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import philox as px
from oracle import ppo_oracle as po
from tests import helpers as H


def test_philox_random123_kat():
    # Random123 kat_vectors, philox4x32-10
    assert [int(v) for v in px.philox4x32_10(0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v) for v in px.philox4x32_10(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)] == [
        0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert [int(v) for v in px.philox4x32_10((0x299F31D0 << 32) | 0xA4093822, 0x243F6A88, 0x85A308D3, 0x13198A2E,
                                             0x03707344)] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_feistel_is_a_permutation():
    for n in (1, 2, 7, 100, 4097):
        p = px.feistel_perm(n, 42, 9)
        assert sorted(p.tolist()) == list(range(n))
    assert not np.array_equal(px.feistel_perm(100, 1, 0), px.feistel_perm(100, 1, 1))


def test_gae_known_answer_vector():
    """minted from ReplayData.compute_returns."""
    g = H.load_golden("gae")
    rewards = np.array([[1, .5], [1, -1], [1, 2], [1, .25]], np.float32).reshape(4, 2, 1, 1)
    vp = np.zeros((5, 2, 1, 1), np.float32)
    vp[:4] = np.array([[.5, .1], [.4, -.2], [.3, .7], [.2, 0]], np.float32).reshape(4, 2, 1, 1)
    masks = np.array([[1, 1], [1, 1], [1, 0], [1, 1], [1, 1]], np.float32).reshape(5, 2, 1, 1)
    bad = np.array([[1, 1], [1, 1], [1, 1], [1, 0], [1, 1]], np.float32).reshape(5, 2, 1, 1)
    nv = np.array([.1, .9], np.float32).reshape(2, 1, 1)
    for proper in (False, True):
        ret, _ = po.compute_returns(rewards, vp, masks, bad, nv, 0.99, 0.95, True, proper, None)
        assert np.array_equal(ret, g["kat_proper%d_returns" % proper])
    ret, _ = po.compute_returns(rewards, vp, masks, bad, nv, 0.99, 0.95, True, False, None)
    np.testing.assert_allclose(ret[:4, 0, 0, 0], [3.7818331718, 2.9367709160, 2.0435094833, 1.0989999771], rtol=1e-7)
    np.testing.assert_allclose(ret[:4, 1, 0, 0], [-0.4504000247, -1.0, 3.0731105804, 1.1410000324], rtol=1e-7)


@pytest.mark.parametrize("use_gae", [True, False])
@pytest.mark.parametrize("proper", [False, True])
@pytest.mark.parametrize("use_vn", [False, True])
def test_gae_random_cases_bit_exact(use_gae, proper, use_vn):
    g = H.load_golden("gae")
    vn = None
    if use_vn:
        vn = po.ValueNormOracle()
        vn.set_state(g["rand_vn_state"])
    ret, vp = po.compute_returns(g["rand_rewards"], g["rand_value_preds"], g["rand_masks"], g["rand_bad_masks"],
                                 g["rand_next_value"], 0.99, 0.95, use_gae, proper, vn)
    tag = "rand_g%d_p%d_v%d" % (use_gae, proper, use_vn)
    assert np.array_equal(ret, g[tag + "_returns"])
    assert np.array_equal(vp, g[tag + "_value_preds"])


def test_minibatch_permutation_is_randperm_chunks():
    g = H.load_golden("perm")
    for key in g:
        _, s, M, n = key.split("_")
        seed, M, nmb = int(s[1:]), int(M[1:]), int(n[1:])
        torch.manual_seed(seed)
        got = np.stack(po.feed_forward_indices(M, nmb))
        assert np.array_equal(got, g[key])
    torch.manual_seed(0)
    assert po.feed_forward_indices(10, 1)[0].tolist() == [4, 1, 7, 5, 3, 9, 0, 8, 6, 2]


@pytest.mark.parametrize("case,seed", [("train_discrete", 0), ("train_discrete_masks", 1), ("train_gaussian", 2)])
def test_init_matches_reference_rng_order(case, seed):
    """PPONet: set_seed(cfg.seed) then policy tower, then critic tower (ppo_net.py:66, ppo_module.py:58-89)."""
    g = H.load_golden(case)
    cfg = H.case_cfg(g)
    pspec, cspec = H.case_specs(g)
    import random
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    tp = po.init_tower(pspec, cfg.gain, cfg.use_orthogonal, cfg.activation_id)
    tc = po.init_tower(cspec, 1.0, cfg.use_orthogonal, cfg.activation_id)
    assert np.array_equal(tp.numpy(), g["theta_p0"])
    assert np.array_equal(tc.numpy(), g["theta_c0"])


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_train_replay_matches_reference(case):
    g = H.load_golden(case)
    r = H.oracle_replay(g)
    np.testing.assert_allclose(r["ptheta"], g["theta_p1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r["ctheta"], g["theta_c1"], rtol=2e-5, atol=2e-6)
    got = np.array([r["info"][k] for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm",
                                           "critic_grad_norm", "ratio")])
    np.testing.assert_allclose(got, g["train_info"], rtol=1e-5, atol=1e-6)
    if "vn_state1" in g:
        np.testing.assert_allclose(r["vn"], g["vn_state1"], rtol=1e-6)


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_update_parity_bar_accepts_the_oracle_and_rejects_broken_updates(case):
    """The bar every train_* parity test applies to d_theta = theta_1 - theta_0 (tests/helpers.py) must be able to FAIL:
    the oracle's replay passes it; an update that skipped its last epoch, an update with one layer left at theta_0
    (= that layer's gradient zeroed) and no update at all are each refused.  (Round-3 VERDICT item 1: the old
    assert_allclose on theta_1 at rtol 2e-3 / atol 3e-5 accepted 'no update' on half of the entries.)"""
    g = H.load_golden(case)
    r = H.oracle_replay(g)
    pblocks, cblocks = H.tower_blocks(r["pspec"]), H.tower_blocks(r["cspec"])
    H.assert_update_parity(g["theta_p0"], r["ptheta"], g["theta_p1"], "policy", blocks=pblocks)
    H.assert_update_parity(g["theta_c0"], r["ctheta"], g["theta_c1"], "critic", blocks=cblocks)
    # a bug confined to a 1..6-entry block (b3, logstd) hides inside the global bar's 1 % exception budget (it ACCEPTS the
    # critic of train_discrete and the policy of train_gaussian without their b3 update); the per-block bar refuses each
    for spec, blocks, th0, th, th1, who in ((r["pspec"], pblocks, g["theta_p0"], r["ptheta"], g["theta_p1"], "policy"),
                                            (r["cspec"], cblocks, g["theta_c0"], r["ctheta"], g["theta_c1"], "critic")):
        for name in ("b3", "logstd"):
            if name in blocks:
                H.assert_update_parity_rejects(th0, H.without_block_update(th0, th, spec, name), th1,
                                               "%s, d%s zeroed" % (who, name), blocks=blocks)
    short = H.oracle_replay(g, skip_epochs=1)
    H.assert_update_parity_rejects(g["theta_p0"], short["ptheta"], g["theta_p1"], "policy, last epoch skipped")
    H.assert_update_parity_rejects(g["theta_c0"], short["ctheta"], g["theta_c1"], "critic, last epoch skipped")
    H.assert_update_parity_rejects(g["theta_p0"], g["theta_p0"], g["theta_p1"], "policy, no update")
    # one layer without its update: every >= 64-entry block of the flat layout (W1, b1, g1, be1, W2, b2, g2, be2, W3)
    for name in ("W1", "b1", "W2", "g2", "W3"):
        broken = H.without_block_update(g["theta_p0"], r["ptheta"], r["pspec"], name)
        H.assert_update_parity_rejects(g["theta_p0"], broken, g["theta_p1"], "policy, d%s zeroed" % name)


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_deterministic_probe_matches_reference(case):
    g = H.load_golden(case)
    pspec, cspec = H.case_specs(g)
    v, a, lp = po.get_actions(pspec, torch.tensor(g["theta_p1"]), cspec, torch.tensor(g["theta_c1"]), g["probe_obs"],
                              g["probe_obs"], g.get("probe_masks"), deterministic=True)
    np.testing.assert_allclose(v, g["probe_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, g["probe_actions"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp, g["probe_logp"], rtol=1e-5, atol=1e-6)


# ---- general towers: oracle/gen_oracle.py pinned on the goldens minted from the reference ---------------------------
GEN_FF_CASES = ["train_gen_h128_l2_tanh_fn", "train_gen_elu_box", "train_gen_leaky_l3", "train_gen_a2c",
                "train_gen_h128_l3_elu_fn"]


def _gen_specs(g):
    from oracle import gen_oracle as go

    cfg = H.case_cfg(g)
    D = g["buf_policy_obs"].shape[-1]
    if "buf_action_masks" in g:
        return cfg, go.specs_from_cfg(cfg, D, g["buf_action_masks"].shape[-1], po.HEAD_CATEGORICAL)
    return cfg, go.specs_from_cfg(cfg, D, g["buf_actions"].shape[-1], po.HEAD_GAUSSIAN)


@pytest.mark.parametrize("case", GEN_FF_CASES)
def test_general_tower_oracle_replays_the_reference(case):
    """MLPBase / MLPLayer of any hidden_size / layer_N / activation / feature norm (mlp.py:8-46,100-180), restated on one
    flat vector incl. the dead ``fc_h`` block: parameter count, the full ``PPOAlgorithm.train`` replay and the
    deterministic probe against the reference's outputs."""
    g = H.load_golden(case)
    cfg, (pspec, cspec) = _gen_specs(g)
    assert pspec.n_params() == g["theta_p0"].size and cspec.n_params() == g["theta_c0"].size
    hp = po.hyper_from_cfg(cfg)
    nmb = cfg.num_mini_batch
    if "a2c" in g:  # A2CAlgorithm: policy-gradient loss, num_mini_batch forced to 1 (a2c.py:37)
        hp.a2c, nmb = True, 1
    ptheta, ctheta = torch.tensor(g["theta_p0"]).clone(), torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    vn = po.ValueNormOracle() if cfg.use_valuenorm else None
    torch.manual_seed(int(g["perm_seed"]))
    info, _, used = po.train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, H.case_buffer(g), cfg.ppo_epoch, nmb)
    assert len(used) == cfg.ppo_epoch * nmb
    np.testing.assert_allclose(ptheta.numpy(), g["theta_p1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ctheta.numpy(), g["theta_c1"], rtol=2e-5, atol=2e-6)
    got = np.array([info[k] for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                      "ratio")])
    np.testing.assert_allclose(got, g["train_info"], rtol=1e-5, atol=1e-6)
    if cfg.layer_N > 1:  # the dead fc_h block never moved
        o = [n for n, _ in pspec.sizes()].index("fc_h_W")
        lo = sum(int(np.prod(s)) for _, s in pspec.sizes()[:o])
        hi = lo + sum(int(np.prod(s)) for _, s in pspec.sizes()[o:o + 4])
        assert np.array_equal(ptheta.numpy()[lo:hi], g["theta_p0"][lo:hi])
    v, a, lp = po.get_actions(pspec, torch.tensor(g["theta_p1"]), cspec, torch.tensor(g["theta_c1"]), g["probe_obs"],
                              g["probe_obs"], g.get("probe_masks"), deterministic=True)
    np.testing.assert_allclose(v, g["probe_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, g["probe_actions"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp, g["probe_logp"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", ["train_share", "train_share_box_fn", "train_share_recurrent", "train_share_h128"])
def test_shared_network_oracle_replays_the_reference(case):
    """PolicyValueNetwork (use_share_model; policy_value_network.py:34-172) restated on one flat vector - obs_prep, common,
    [GRU], v_out, act - with the reference's two clips of the same gradient (ppo.py:126-145): full train replay."""
    from oracle import gen_oracle as go

    g = H.load_golden(case)
    cfg = H.case_cfg(g)
    D = g["buf_policy_obs"].shape[-1]
    if "buf_action_masks" in g:
        aspec, cspec = go.shared_specs_from_cfg(cfg, D, g["buf_action_masks"].shape[-1], po.HEAD_CATEGORICAL)
    else:
        aspec, cspec = go.shared_specs_from_cfg(cfg, D, g["buf_actions"].shape[-1], po.HEAD_GAUSSIAN)
    assert aspec.n_params() == g["theta_m0"].size
    hp = po.hyper_from_cfg(cfg)
    theta = torch.tensor(g["theta_m0"]).clone()
    adam = po.AdamOracle(theta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    vn = po.ValueNormOracle() if cfg.use_valuenorm else None
    torch.manual_seed(int(g["perm_seed"]))
    rec = bool(cfg.use_recurrent_policy)
    info = go.train_shared(hp, aspec, cspec, theta, adam, vn, H.case_buffer(g), cfg.ppo_epoch, cfg.num_mini_batch,
                           cfg.data_chunk_length if rec else 0)
    np.testing.assert_allclose(theta.numpy(), g["theta_m1"], rtol=3e-5, atol=3e-6)
    got = np.array([info[k] for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                      "ratio")])
    np.testing.assert_allclose(got, g["train_info"], rtol=2e-5, atol=1e-6)
    if vn is not None:
        np.testing.assert_allclose(vn.state(), g["vn_state1"], rtol=1e-6)


def test_multidiscrete_actlayer_oracle_matches_the_reference():
    """ACTLayer with MultiDiscrete([3, 2, 5]) (act.py:26-34,60-72,136-151): log-probs, the detached entropy, the gradient
    of sum(log-probs) w.r.t. features and head parameters, and the deterministic actions."""
    from oracle import gen_oracle as go

    g = H.load_golden("actlayer_multidiscrete")
    n = len(g["nvec"])
    t = lambda a: torch.tensor(a)
    x = t(g["x"]).clone().requires_grad_(True)
    Ws = [t(g["W%d" % i]).clone().requires_grad_(True) for i in range(n)]
    bs = [t(g["b%d" % i]).clone().requires_grad_(True) for i in range(n)]
    logp, ent = go.multidiscrete_evaluate(x, Ws, bs, t(g["actions"]), t(g["active"]))
    logp.sum().backward()
    np.testing.assert_allclose(logp.detach().numpy(), g["logp"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(float(ent), float(g["entropy"]), rtol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-5, atol=1e-6)
    for i in range(n):
        np.testing.assert_allclose(Ws[i].grad.numpy(), g["dW%d" % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bs[i].grad.numpy(), g["db%d" % i], rtol=1e-5, atol=1e-6)
    a, lp = go.multidiscrete_mode(t(g["x"]), [w.detach() for w in Ws], [b.detach() for b in bs])
    assert np.array_equal(a.numpy(), g["det_actions"])
    np.testing.assert_allclose(lp.numpy(), g["det_logp"], rtol=1e-6, atol=1e-6)


def _oracle_full_general(name):
    """Replay a ``train_cfg{3,5}_full`` golden (feed-forward, full BASELINE size) with the oracle restatement."""
    from oracle.fixtures import synth_update_buffer_general

    g = H.load_golden(name)
    N, T, Dp, Dc, n_act, A, seed, legal, rec = (int(x) for x in g["shape"])
    kind = str(g["kind"])
    assert not rec and Dp == Dc
    cfg = H.case_cfg(g)
    hp = po.hyper_from_cfg(cfg)
    buf = synth_update_buffer_general(seed, N, T, Dp, Dc, kind, n_act, A, bool(legal), 0)
    vn = po.ValueNormOracle()
    nv = buf.pop("next_value")
    buf["returns"], buf["value_preds"] = po.compute_returns(buf["rewards"], buf["value_preds"], buf["masks"],
                                                            buf["bad_masks"], nv, cfg.gamma, cfg.gae_lambda,
                                                            value_normalizer=vn)
    buf.setdefault("action_masks", None)
    probe = np.array([buf["returns"][t, n, a, 0] for t, n, a in g["returns_probe_idx"]])
    np.testing.assert_array_equal(probe, g["returns_probe"])
    pspec = po.TowerSpec(Dp, n_act, po.HEAD_CATEGORICAL if kind == "discrete" else po.HEAD_GAUSSIAN)
    cspec = po.TowerSpec(Dc, 1, po.HEAD_VALUE)
    ptheta, ctheta = torch.tensor(g["theta_p0"]).clone(), torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    torch.manual_seed(int(g["perm_seed"]))
    info, _, _ = po.train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf, cfg.ppo_epoch, cfg.num_mini_batch)
    keys = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")
    np.testing.assert_allclose([info[k] for k in keys], g["train_info"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(ptheta.numpy(), g["theta_p1"], rtol=3e-4, atol=3e-6)
    np.testing.assert_allclose(ctheta.numpy(), g["theta_c1"], rtol=3e-4, atol=3e-6)
    np.testing.assert_allclose(vn.state(), g["vn_state1"], rtol=1e-5)


def test_oracle_replays_the_full_size_cfg3_reference_update():
    """BASELINE.json configs[2] at full size (1024 x 200 rows, obs 17, Box(6), 10 epochs): the oracle restatement lands on
    the REAL reference's outputs (per-dimension Gaussian log-probs, FixedNormal.log_probs) - ~10 s of torch CPU."""
    _oracle_full_general("train_cfg3_full")


def test_mixed_actlayer_oracle_matches_the_reference():
    """ACTLayer with Tuple(Box(2), Discrete(5)) - the mixed branch (act.py:33-63,126-147): joint log-prob, the
    0.0025 / 0.01-weighted entropy with and without active masks, the gradient of sum(log-prob) + 3 * entropy w.r.t.
    features and head parameters, and the deterministic actions."""
    from oracle import gen_oracle as go

    g = H.load_golden("actlayer_mixed")
    t = lambda a: torch.tensor(a)
    x = t(g["x"]).clone().requires_grad_(True)
    P = {k: t(g[k]).clone().requires_grad_(True) for k in ("Wm", "bm", "logstd", "Wc", "bc")}
    logp, ent = go.mixed_evaluate(x, P["Wm"], P["bm"], P["logstd"], P["Wc"], P["bc"], t(g["actions"]), t(g["active"]))
    (logp.sum() + 3.0 * ent).backward()
    np.testing.assert_allclose(logp.detach().numpy(), g["logp"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(float(ent), float(g["entropy"]), rtol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-5, atol=1e-6)
    for k in P:
        np.testing.assert_allclose(P[k].grad.numpy(), g["d" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        _, ent0 = go.mixed_evaluate(x, P["Wm"], P["bm"], P["logstd"], P["Wc"], P["bc"], t(g["actions"]), None)
        a, lp = go.mixed_mode(x, P["Wm"], P["bm"], P["logstd"], P["Wc"], P["bc"])
    np.testing.assert_allclose(float(ent0), float(g["entropy_nomask"]), rtol=1e-6)
    np.testing.assert_allclose(a.numpy(), g["det_actions"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), g["det_logp"], rtol=1e-6, atol=1e-6)


def test_oracle_replays_the_full_size_cfg5_reference_update():
    """BASELINE.json configs[4] at full size (4096 x 200 rows, obs 18, Discrete(9) with random legal-move masks, 10
    epochs): the oracle restatement lands on the REAL reference's outputs - ~40 s of torch CPU."""
    _oracle_full_general("train_cfg5_full")

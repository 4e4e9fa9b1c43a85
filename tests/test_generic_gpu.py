"""General towers (modules/generic_net.py + csrc/orl_gen.hip) against the REAL reference: hidden_size / layer_N /
activation_id / use_feature_normalization outside the fused default tower, the shared PolicyValueNetwork, and the
MultiDiscrete ACTLayer.  Golden vectors: oracle/gen_golden.py (reference mlp.py:8-46,100-180,
policy_value_network.py:34-230, act.py:14-172, algorithms/ppo.py:46-176).  Needs a MI355X."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GEN_CASES = ["train_gen_h128_l2_tanh_fn", "train_gen_elu_box", "train_gen_leaky_l3", "train_share", "train_share_box_fn",
             "train_share_h128", "train_gen_h128_l3_elu_fn",
             "train_gen_a2c", "train_gen_mixed"]
# fp32 tolerances of the update path (MFMA k-order, tile-wise gradient sums); same as tests/test_ppo_update_gpu.py
THETA_RTOL, THETA_ATOL = 2e-3, 3e-5
INFO_RTOL, INFO_ATOL = 3e-4, 3e-5


def build(g, load_theta0=True):
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule
    from openrl_amd.utils.util import set_seed

    cfg = H.case_cfg(g)
    cfg.seed = int(g["perm_seed"]) - 1234  # gen_golden._train_case: cfg.seed = seed, perm_seed = 1234 + seed
    T, N, D = g["buf_policy_obs"].shape[0] - 1, g["buf_policy_obs"].shape[1], g["buf_policy_obs"].shape[-1]
    cfg.episode_length, cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = T, N, 1, cfg.hidden_size
    obs_space = spaces.Box(-np.inf, np.inf, (D,))
    if "mixed" in g:  # the reference's mixed branch: Tuple(Box(cd), Discrete(n))
        act_space = spaces.Tuple((spaces.Box(-1, 1, (int(g["mixed"][0]),)), spaces.Discrete(int(g["mixed"][1]))))
    elif "buf_action_masks" in g:
        act_space = spaces.Discrete(g["buf_action_masks"].shape[-1])
    else:
        act_space = spaces.Box(-1, 1, (g["buf_actions"].shape[-1],))
    set_seed(cfg.seed)
    module = PPOModule(cfg, obs_space, obs_space, act_space, share_model=cfg.use_share_model, device=DEV, rank=0,
                       world_size=1)
    assert module.generic, "these configurations must take the general path"
    return cfg, module, obs_space, act_space, ReplayData, PPOAlgorithm


def nets(module, g):
    if "theta_m0" in g:
        return [("model", "theta_m0", "theta_m1")]
    return [("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")]


@pytest.mark.parametrize("case", GEN_CASES)
def test_initial_weights_reproduce_the_reference(case):
    """Same seed -> same generator consumption as PolicyNetwork / ValueNetwork / PolicyValueNetwork construction
    (feature norm, fc1, fc_h + its clones, fc3, common, v_out, act heads), including the never-used fc_h block.
    Bit-exact on the host that generated the goldens (tests/test_generic_cpu.py); another host's LAPACK QR inside
    orthogonal_ differs in the last bits, hence the 1e-5 here."""
    g = H.load_golden(case)
    cfg, module, *_ = build(g)
    for name, k0, _ in nets(module, g):
        got = module.models[name].reference_flat().cpu().numpy()
        assert got.shape == g[k0].shape
        np.testing.assert_allclose(got, g[k0], rtol=0, atol=1e-5, err_msg=name)


@pytest.mark.parametrize("case", ["train_gen_h128_l2_tanh_fn", "train_share"])
def test_update_parity_bar_rejects_a_short_general_update(case):
    """Negative control of the d_theta bar on the general towers: the golden case with its last epoch skipped must be
    refused (and 'no update' as well)."""
    g = H.load_golden(case)
    cfg, module, obs_space, act_space, ReplayData, PPOAlgorithm = build(g)
    for name, k0, _ in nets(module, g):
        module.models[name].load_reference_flat(g[k0])
    buf = ReplayData(cfg, 1, obs_space, act_space, device=DEV)
    for f in ("policy_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "bad_masks",
              "active_masks", "action_masks"):
        if "buf_" + f in g:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    algo = PPOAlgorithm(cfg, module, agent_num=1, device=DEV)
    algo.ppo_epoch -= 1
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    algo.train(buf)
    for name, k0, k1 in nets(module, g):
        H.assert_update_parity_rejects(g[k0], module.models[name].reference_flat().cpu().numpy(), g[k1],
                                       name + ", last epoch skipped")
        H.assert_update_parity_rejects(g[k0], g[k0], g[k1], name + ", no update")


@pytest.mark.parametrize("case", GEN_CASES)
def test_train_matches_reference_golden(case):
    g = H.load_golden(case)
    cfg, module, obs_space, act_space, ReplayData, PPOAlgorithm = build(g)
    for name, k0, _ in nets(module, g):
        module.models[name].load_reference_flat(g[k0])
    buf = ReplayData(cfg, 1, obs_space, act_space, device=DEV)
    for f in ("policy_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "bad_masks",
              "active_masks", "action_masks"):
        if "buf_" + f in g:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    if "a2c" in g:  # A2CAlgorithm: policy-gradient loss, one minibatch per epoch, no ratio (a2c.py:27-145)
        from openrl_amd.algorithms.a2c import A2CAlgorithm as PPOAlgorithm  # noqa: F811
    algo = PPOAlgorithm(cfg, module, agent_num=1, device=DEV)
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    got = np.array([info.get(k, 0.0) for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm",
                                               "critic_grad_norm", "ratio")])
    np.testing.assert_allclose(got, g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
    for name, k0, k1 in nets(module, g):
        got_flat = module.models[name].reference_flat().cpu().numpy()
        H.assert_update_parity(g[k0], got_flat, g[k1], name)  # the bar on d_theta (tests/helpers.py)
        np.testing.assert_allclose(got_flat, g[k1], rtol=THETA_RTOL, atol=THETA_ATOL, err_msg=name)
    if "vn_state1" in g:
        np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"],
                                   rtol=1e-5)
    # deterministic probe on the trained weights: values, greedy / mean actions, their log-probs
    module2 = module
    for name, _, k1 in nets(module, g):
        module2.models[name].load_reference_flat(g[k1])
    pm = g["probe_masks"] if "probe_masks" in g else None
    v, a, lp, _, _ = module2.get_actions(g["probe_obs"], g["probe_obs"], None, None, None, action_masks=pm,
                                         deterministic=True)
    np.testing.assert_allclose(v.cpu().numpy(), g["probe_values"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(a.cpu().numpy(), g["probe_actions"], rtol=2e-4, atol=2e-5)
    lp = lp.cpu().numpy()
    if "mixed" in g:  # ONE joint log-prob [B, 1] in the reference, replicated over the stored columns here
        assert np.all(lp == lp[:, :1])
        lp = lp[:, :1]
    np.testing.assert_allclose(lp, g["probe_logp"], rtol=2e-4, atol=2e-5)


def test_dead_fc_h_block_never_moves():
    """MLPLayer registers fc_h next to its clones fc2 and never runs it (mlp.py:28-34,41-46): no gradient, no Adam
    step - its parameters stay at their initial values through an update, as in the reference golden."""
    g = H.load_golden("train_gen_leaky_l3")
    cfg, module, *_ = build(g)
    net = module.models["policy"]
    keys = [k for k, _, _ in net.entries if ".fc_h." in k]
    assert keys, "layer_N = 3 must register an fc_h block"
    flat0, flat1 = g["theta_p0"], g["theta_p1"]
    o = 0
    for k, t in net.named_parameters():
        n = t.numel()
        if ".fc_h." in k:
            assert np.array_equal(flat0[o:o + n], flat1[o:o + n]), k  # the reference itself never moves it
        o += n


def test_mixed_actlayer_vs_reference():
    """ACTLayer with Tuple(Box(2), Discrete(5)) - the reference's mixed branch (act.py:33-63, 126-147) - through the
    general path's head kernels: joint log-prob in every stored column, the 0.0025 / 0.01-weighted entropy with and
    without active masks, deterministic actions (Gaussian mean + argmax) with their joint log-prob."""
    from openrl_amd import ops, ops_gen, spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules import generic_net as gn

    g = H.load_golden("actlayer_mixed")
    cd, n = (int(v) for v in g["shape"])
    net = gn.GenNet("policy", default_cfg([]), 4, spaces.Tuple((spaces.Box(-1, 1, (cd,)), spaces.Discrete(n))), DEV)
    sd = {"act.action_outs.0.fc_mean.weight": g["Wm"], "act.action_outs.0.fc_mean.bias": g["bm"],
          "act.action_outs.0.logstd._bias": g["logstd"], "act.action_outs.1.linear.weight": g["Wc"],
          "act.action_outs.1.linear.bias": g["bc"]}
    seen = set()
    for k, t in net.named_parameters():
        if k in sd:
            t.copy_(torch.tensor(sd[k]).reshape(t.shape))
            seen.add(k)
    assert seen == set(sd), "the mixed head's state_dict keys are the reference's"
    B = g["x"].shape[0]
    feats = torch.tensor(g["x"], device=DEV)
    ws = gn.GenWorkspace(net, B, True)
    logits = gn.head_forward(net, ws, "act", feats)
    a_w = cd + 1
    logstd = net.v(net.heads["act"]["logstd"], cd)
    R = ops.record_width(4, 4, a_w, 0)
    for masks in (True, False):
        cfg = default_cfg([])
        cfg.use_policy_active_masks = masks
        rec = torch.zeros(B, R, device=DEV)
        rec[:, 8:8 + a_w] = torch.tensor(g["actions"], device=DEV)
        rec[:, 8 + 2 * a_w + 3] = torch.tensor(g["active"][:, 0], device=DEV)
        logp, ent = torch.empty(B, a_w, device=DEV), torch.empty(B, device=DEV)
        ops_gen.policy_eval(net.head_desc, logits, logstd, rec, 4, 4, a_w, 0, B, ops.make_hparams(cfg), logp, ent)
        for k in range(a_w):  # ONE joint log-prob, replicated over the stored columns
            np.testing.assert_allclose(logp[:, k:k + 1].cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
        if masks:
            act = torch.tensor(g["active"][:, 0], device=DEV)
            np.testing.assert_allclose(float((ent * act).sum() / act.sum()), float(g["entropy"]), rtol=1e-5)
        else:
            np.testing.assert_allclose(float(ent.mean()), float(g["entropy_nomask"]), rtol=1e-5)
    a = torch.empty(B, a_w, device=DEV)
    lp = torch.empty(B, a_w, device=DEV)
    ops_gen.sample(net.head_desc, logits, logstd, None, B, True, 0, 0, 0, None, None, a_w, a, lp)
    np.testing.assert_allclose(a.cpu().numpy(), g["det_actions"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp[:, :1].cpu().numpy(), g["det_logp"], rtol=1e-5, atol=1e-6)
    # stochastic draws under forced deviates: a = mean + std * eps, class = inverse CDF of the forced uniform
    forced = torch.tensor(np.concatenate([g["actions"][:, :cd] * 0 + 0.5, np.full((B, 1), 0.999, np.float32)], 1), device=DEV)
    ops_gen.sample(net.head_desc, logits, logstd, None, B, False, 0, 0, 0, None, forced, a_w, a, lp)
    mean = logits[:, :cd]
    np.testing.assert_allclose(a[:, :cd].cpu().numpy(), (mean + logstd.exp() * 0.5).cpu().numpy(), rtol=1e-6, atol=1e-6)
    cum = torch.softmax(logits[:, cd:].double(), -1).cumsum(-1)
    want = (cum > 0.999 * cum[:, -1:]).float().argmax(-1)
    assert (a[:, cd].long() == want).float().mean() >= 0.95  # a CDF edge within fp32 rounding of the uniform may flip


def test_multidiscrete_actlayer_vs_reference():
    """ACTLayer with MultiDiscrete([3, 2, 5]) (act.py:26-34,60-72,136-151): log-probs of given actions, the detached
    mean entropy, greedy actions, and the gradient of sum(log-probs) through the head."""
    from openrl_amd import ops, ops_gen, spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules import generic_net as gn

    g = H.load_golden("actlayer_multidiscrete")
    nvec = [int(k) for k in g["nvec"]]
    cfg = default_cfg([])
    net = gn.GenNet("policy", cfg, 4, spaces.MultiDiscrete(nvec), DEV)
    sd = {"act.action_outs.%d.linear.weight" % i: g["W%d" % i] for i in range(len(nvec))}
    sd.update({"act.action_outs.%d.linear.bias" % i: g["b%d" % i] for i in range(len(nvec))})
    for k, t in net.named_parameters():
        if k in sd:
            t.copy_(torch.tensor(sd[k]).reshape(t.shape))
    B = g["x"].shape[0]
    feats = torch.tensor(g["x"], device=DEV)
    ws = gn.GenWorkspace(net, B, True)
    logits = gn.head_forward(net, ws, "act", feats)
    a_w, n_tot = len(nvec), sum(nvec)
    R = ops.record_width(4, 4, a_w, 0)
    rec = torch.zeros(B, R, device=DEV)
    rec[:, 8:8 + a_w] = torch.tensor(g["actions"], device=DEV)
    rec[:, 8 + 2 * a_w + 3] = torch.tensor(g["active"][:, 0], device=DEV)
    logp, ent = torch.empty(B, a_w, device=DEV), torch.empty(B, device=DEV)
    hp = ops.make_hparams(cfg)
    ops_gen.policy_eval(net.head_desc, logits, None, rec, 4, 4, a_w, 0, B, hp, logp, ent)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    act = torch.tensor(g["active"][:, 0], device=DEV)
    np.testing.assert_allclose(float((ent * act).sum() / act.sum()), float(g["entropy"]), rtol=1e-5)
    # greedy actions + their log-probs
    a = torch.empty(B, a_w, device=DEV)
    lp = torch.empty(B, a_w, device=DEV)
    ops_gen.sample(net.head_desc, logits, None, None, B, True, 0, 0, 0, None, None, a_w, a, lp)
    assert np.array_equal(a.cpu().numpy(), g["det_actions"])
    np.testing.assert_allclose(lp.cpu().numpy(), g["det_logp"], rtol=1e-5, atol=1e-6)
    # gradient of sum(logp) w.r.t. features and head parameters: d logp_h / d logits = onehot - softmax
    dl = torch.zeros(B, n_tot, device=DEV)
    off = 0
    for h, k in enumerate(nvec):
        p = torch.softmax(logits[:, off:off + k], -1)
        dl[:, off:off + k] = -p
        dl[torch.arange(B), off + torch.tensor(g["actions"][:, h], device=DEV).long()] += 1.0
        off += k
    dfeat = torch.empty(B, net.H, device=DEV)
    net.grad.zero_()
    gn.head_backward(net, ws, "act", feats, dl.contiguous(), dfeat, False)
    np.testing.assert_allclose(dfeat.cpu().numpy(), g["dx"], rtol=1e-4, atol=1e-6)
    grads = {k: net.v(o, *s, grad=True).cpu().numpy() for k, s, o in net.entries}
    for i in range(len(nvec)):
        np.testing.assert_allclose(grads["act.action_outs.%d.linear.weight" % i], g["dW%d" % i], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(grads["act.action_outs.%d.linear.bias" % i], g["db%d" % i], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,N,K,split", [(70, 33, 50, 1), (64, 64, 16, 1), (5, 130, 1000, 4), (128, 6, 4096, 8),
                                           (1000, 48, 48, 1)])
def test_gemm_against_torch_fp32(M, N, K, split):
    """orl_gemm (fp32 MFMA, arbitrary strides, split-K) vs a plain torch fp32 reference of the same product."""
    from openrl_amd import ops_gen

    rs = np.random.RandomState(M + N + K)
    A, Bm = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    want = torch.tensor(A, dtype=torch.float64) @ torch.tensor(Bm, dtype=torch.float64)
    a, b = torch.tensor(A, device=DEV), torch.tensor(Bm, device=DEV)
    part = torch.empty(split * M * N, device=DEV)
    for at in (False, True):        # A stored [M,K] or transposed [K,M]
        for bt in (False, True):    # B stored [K,N] or transposed [N,K]
            aa = a.t().contiguous() if at else a
            bb = b.t().contiguous() if bt else b
            c = torch.zeros(M, N, device=DEV)
            ops_gen.gemm(aa, 1 if at else K, M if at else 1, bb, 1 if bt else N, K if bt else 1, c, N, M, N, K, split, part)
            np.testing.assert_allclose(c.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-4 * np.sqrt(K))


_ACTS = {0: torch.tanh, 1: torch.relu, 2: torch.nn.functional.leaky_relu, 3: torch.nn.functional.elu}


@pytest.mark.parametrize("B,n_in,n_out,act,ln", [(1000, 4, 128, 1, True), (777, 128, 128, 0, True), (300, 20, 36, 3, True),
                                                 (530, 256, 256, 2, True), (200, 512, 512, 1, True), (999, 64, 64, -1, True),
                                                 (640, 128, 5, -1, False), (333, 6, 100, 1, True), (64, 128, 1, -1, False),
                                                 (20001, 128, 128, 1, True), (17000, 4, 64, 0, True), (16500, 256, 256, 1, True),
                                                 (16400, 64, 9, -1, False), (18000, 32, 32, 2, True), (17001, 96, 96, 3, True),
                                                 (16390, 20, 128, 1, True), (700, 64, 384, -1, False), (17000, 128, 384, -1, False), (900, 36, 300, 0, False)])
def test_fused_layer_kernels_against_torch_fp64(B, n_in, n_out, act, ln):
    """orl_gen_layer_fwd / orl_gen_layer_bwd / orl_gen_wgrad / orl_gen_colsum vs nn.Sequential(Linear, act, LayerNorm)
    evaluated by torch autograd in fp64 (mlp.py:8-46).  Tolerances: fp32 accumulation over K <= 512 products and B rows."""
    from openrl_amd import ops_gen

    rs = np.random.RandomState(B + n_in + n_out)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    x, W, b = t64(rs.randn(B, n_in)), t64(rs.randn(n_out, n_in) / np.sqrt(n_in)), t64(0.1 * rs.randn(n_out))
    g, be = t64(1.0 + 0.1 * rs.randn(n_out)), t64(0.1 * rs.randn(n_out))
    dy = torch.tensor(rs.randn(B, n_out), dtype=torch.float64)
    a_ref = x @ W.t() + b
    if act >= 0:
        a_ref = _ACTS[act](a_ref)
    y_ref = torch.nn.functional.layer_norm(a_ref, (n_out,), g, be, 1e-5) if ln else a_ref
    (y_ref * dy).sum().backward()

    dev = lambda t: t.detach().to(torch.float32).to(DEV).contiguous()
    # the weights sit at an arbitrary (possibly 16-byte misaligned) offset of a flat vector, like GenNet.theta
    flat = torch.zeros(3 + n_out * n_in, device=DEV)
    flat[3:] = dev(W).view(-1)
    for Wd in (dev(W), flat[3:].view(n_out, n_in)):
        xd, a_out, stats, y = dev(x), torch.zeros(B, n_out, device=DEV), torch.zeros(B, 2, device=DEV), torch.zeros(B, n_out, device=DEV)
        act_id = ops_gen.ACT_NONE if act < 0 else act
        ops_gen.layer_fwd(xd, Wd, dev(b), act_id, dev(g) if ln else None, dev(be) if ln else None, a_out, stats if ln else None, y)
        np.testing.assert_allclose(a_out.cpu().numpy(), a_ref.detach().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), rtol=1e-4, atol=1e-4)
        # inference form: no a / stats
        y2 = torch.zeros_like(y)
        ops_gen.layer_fwd(xd, Wd, dev(b), act_id, dev(g) if ln else None, dev(be) if ln else None, None, None, y2)
        assert torch.equal(y, y2)

        part = torch.zeros(ops_gen.MAX_BLOCKS * 3 * n_out, device=DEV)
        dz, dx = torch.zeros(B, n_out, device=DEV), torch.zeros(B, n_in, device=DEV)
        square = n_in == n_out
        nb = ops_gen.layer_bwd(dev(dy), a_out, stats if ln else None, dev(g) if ln else None, act_id,
                               Wd if square else None, dz, dx if square else None, part)
        dg, dbe, db = torch.zeros(n_out, device=DEV), torch.zeros(n_out, device=DEV), torch.zeros(n_out, device=DEV)
        ops_gen.colsum(part, nb, [(dg if ln else None, n_out), (dbe if ln else None, n_out), (db, n_out)])
        sB = np.sqrt(B)
        np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=2e-5 * sB)
        if ln:
            np.testing.assert_allclose(dg.cpu().numpy(), g.grad.numpy(), rtol=1e-4, atol=2e-5 * sB)
            np.testing.assert_allclose(dbe.cpu().numpy(), be.grad.numpy(), rtol=1e-4, atol=2e-5 * sB)
        if not square:
            ops_gen.linear_dgrad(dz, Wd.contiguous(), dx)
        np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-4)
        dW = torch.zeros(n_out, n_in, device=DEV)
        wp = torch.zeros(max(n_out * n_in, min(512 * n_out * n_in, 1 << 22)), device=DEV)
        ops_gen.wgrad(dz, xd, dW, wp)
        np.testing.assert_allclose(dW.cpu().numpy(), W.grad.numpy(), rtol=1e-4, atol=2e-5 * sB)
        # deterministic: a second run is bit-identical
        dW2 = torch.zeros_like(dW)
        ops_gen.wgrad(dz, xd, dW2, wp)
        assert torch.equal(dW, dW2)


@pytest.mark.parametrize("argv,D,B", [(["--hidden_size", "128"], 4, 4096), (["--hidden_size", "128", "--layer_N", "4", "--activation_id", "0"], 18, 1000),
                                      (["--hidden_size", "36", "--layer_N", "2", "--use_feature_normalization", "true", "--activation_id", "3"], 7, 333),
                                      (["--hidden_size", "256", "--layer_N", "2", "--activation_id", "2"], 54, 77),
                                      (["--hidden_size", "64", "--use_share_model", "true", "--layer_N", "2"], 6, 500)])
def test_whole_tower_launch_equals_the_layer_by_layer_forward(argv, D, B):
    """orl_gen_mlp_fwd (the rollout's one-launch tower) against the per-layer kernels the update runs (which are checked
    against torch above): logits / values agree to fp32 summation order (the K index is walked in a permuted order)."""
    from openrl_amd import spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules import generic_net as gn
    from openrl_amd import ops_gen

    cfg = default_cfg(argv)
    share = bool(cfg.use_share_model)
    mod = gn.GenericPPOModule(cfg, spaces.Box(-np.inf, np.inf, (D,)), spaces.Box(-np.inf, np.inf, (D,)), spaces.Discrete(5),
                              share_model=share, device=DEV)
    torch.manual_seed(B)
    x = torch.randn(B, D, device=DEV)
    for net, heads in ((mod.policy_net, ("act", "v_out") if share else ("act",)), (mod.critic_net, ("v_out",))):
        net.theta.add_(0.05 * torch.randn_like(net.theta))  # biases / LayerNorm affine away from their 0 / 1 init
        desc = net.mlp_desc(heads)
        assert desc is not None
        outs = [torch.zeros(B, net.heads[h]["n"], device=DEV) for h in heads]
        ops_gen.mlp_fwd(desc, x, outs[0], outs[1] if len(outs) > 1 else None)
        ws = gn.GenWorkspace(net, B, False)
        feats = gn.trunk_forward(net, ws, x, False)
        for h, got in zip(heads, outs):
            want = gn.head_forward(net, ws, h, feats)
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("argv,D,Dc,space,B", [
    (["--hidden_size", "128"], 4, 4, "disc", 4096),
    (["--hidden_size", "128", "--layer_N", "4", "--activation_id", "0"], 18, 54, "disc", 1000),
    (["--hidden_size", "36", "--layer_N", "2", "--use_feature_normalization", "true", "--activation_id", "3"], 7, 7, "box", 333),
    (["--hidden_size", "256", "--layer_N", "2", "--activation_id", "2"], 54, 20, "md", 77),
    (["--hidden_size", "64", "--use_share_model", "true", "--layer_N", "2"], 6, 6, "disc", 500),
    (["--hidden_size", "64", "--use_share_model", "true"], 6, 6, "box", 17)])
@pytest.mark.parametrize("deterministic", [False, True])
def test_one_launch_act_step_is_bit_identical_to_the_three_launch_route(argv, D, Dc, space, B, deterministic):
    """orl_gen_act (policy tower + ACTLayer sampling + critic tower / shared value head in one launch) against
    orl_gen_mlp_fwd + orl_gen_sample + orl_gen_mlp_fwd: the same MFMA order and the same Philox counters, so actions,
    log-probs and values are equal bit for bit; a device step counter and action masks are covered too."""
    from openrl_amd import spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules import generic_net as gn
    from openrl_amd import ops_gen

    cfg = default_cfg(argv)
    share = bool(cfg.use_share_model)
    asp = {"disc": spaces.Discrete(5), "box": spaces.Box(-1.0, 1.0, (6,)), "md": spaces.MultiDiscrete([3, 4, 2])}[space]
    mod = gn.GenericPPOModule(cfg, spaces.Box(-np.inf, np.inf, (D,)), spaces.Box(-np.inf, np.inf, (Dc,)), asp,
                              share_model=share, device=DEV)
    torch.manual_seed(B)
    pn, cn = mod.policy_net, mod.critic_net
    for net in ({id(pn): pn, id(cn): cn}).values():
        net.theta.add_(0.05 * torch.randn_like(net.theta))
    x, xc = torch.randn(B, D, device=DEV), torch.randn(B, Dc, device=DEV)
    am = None
    if space == "disc":
        am = (torch.rand(B, 5, device=DEV) > 0.3).float()
        am[:, 2] = 1.0
    a_w, nl = mod.act_width, pn.heads["act"]["n"]
    step_dev = torch.tensor([12345], dtype=torch.int64, device=DEV)
    f = lambda *sh: torch.full(sh, 7.0, device=DEV)
    # --- three launches
    logits, v_ref, a_ref, lp_ref = f(B, nl), f(B, 1), f(B, a_w), f(B, a_w)
    if share:
        ops_gen.mlp_fwd(pn.mlp_desc(("act", "v_out")), x, logits, v_ref)
    else:
        ops_gen.mlp_fwd(pn.mlp_desc(("act",)), x, logits)
        ops_gen.mlp_fwd(cn.mlp_desc(("v_out",)), xc, v_ref)
    ops_gen.sample(pn.head_desc, logits, mod._logstd(), am, B, deterministic, 11, 3, 99, step_dev, None, a_w, a_ref, lp_ref)
    # --- one launch
    lg2, v, a, lp = f(B, nl), f(B, 1), f(B, a_w), f(B, a_w)
    ops_gen.act_step(pn.mlp_desc(("act", "v_out")) if share else pn.mlp_desc(("act",)), x,
                     None if share else cn.mlp_desc(("v_out",)), None if share else xc, v, pn.head_desc, mod._logstd(), am,
                     deterministic, 11, 3, 99, step_dev, None, a_w, a, lp, logits_out=lg2)
    assert torch.equal(lg2, logits) and torch.equal(v, v_ref)
    assert torch.equal(a, a_ref) and torch.equal(lp, lp_ref)
    if space == "disc":
        assert bool((am.gather(1, a.long()) == 1).all())
    # --- the module's rollout entry takes the one-launch route and draws what the three launches draw
    mod.act_seed, mod.rng_step, mod.rng_step_dev = 11, 99, step_dev
    vals, acts, lps = mod._forward(x if share else xc, x, am, deterministic)
    a_ref2, lp_ref2 = f(B, a_w), f(B, a_w)
    ops_gen.sample(pn.head_desc, logits, mod._logstd(), am, B, deterministic, 11, 0, 99, step_dev, None, a_w, a_ref2, lp_ref2)
    assert torch.equal(acts, a_ref2) and torch.equal(lps, lp_ref2) and torch.equal(vals, v_ref)
    assert mod.rng_step == (99 if deterministic else 100)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
@pytest.mark.parametrize("H,L,N,rN", [(128, 5, 70, 1), (36, 3, 200, 1), (64, 10, 33, 1), (168, 2, 50, 1), (256, 3, 40, 1),
                                       (512, 2, 20, 1), (64, 4, 60, 2), (32, 3, 50, 3)])
def test_gru_sequence_forward_backward_against_torch_fp64(H, L, N, rN, cell):
    """generic_net.gru_forward / gru_backward (projection GEMMs + orl_gen_gru_gate_fwd/_bwd + the LayerNorm after the
    stack) vs a stack of GRU cells + LayerNorm under torch autograd in fp64, with the reference's masking
    h_{t-1} * mask_t on every layer (networks/utils/rnn.py:39-99); recurrent_N = 1, 2, 3 layers."""
    from openrl_amd import spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules import generic_net as gn

    cfg = default_cfg(["--hidden_size", str(H), "--use_recurrent_policy", "true", "--recurrent_N", str(rN), "--rnn_type", cell])
    net = gn.GenNet("policy", cfg, 6, spaces.Discrete(3), DEV, recurrent=True)
    G, SW = net.G, net.state_w
    assert (G, SW) == ((3, H) if cell == "gru" else (4, 2 * H))
    torch.manual_seed(H + L)
    net.host_init(cfg)
    net.theta.add_(0.05 * torch.randn_like(net.theta))
    r = net.rnn
    t64 = lambda off, *sh: net.v(off, *sh).detach().cpu().double().clone().requires_grad_(True)
    P = [dict(Wih=t64(ly["Wih"], G * H, H), Whh=t64(ly["Whh"], G * H, H), bih=t64(ly["bih"], G * H), bhh=t64(ly["bhh"], G * H))
         for ly in r["layers"]]
    g, be = t64(r["g"], H), t64(r["be"], H)
    feats = torch.randn(L * N, H, dtype=torch.float64, requires_grad=True)
    h0 = torch.randn(N, rN, SW, dtype=torch.float64)
    masks = (torch.rand(L * N) > 0.2).double()
    dy = torch.randn(L * N, H, dtype=torch.float64)
    hs, ys = [h0[:, k] for k in range(rN)], []   # per layer: h (GRU) or [h | c] (LSTM)
    for t in range(L):
        s = slice(t * N, (t + 1) * N)
        x = feats[s]
        for k, p in enumerate(P):
            st = hs[k] * masks[s, None]
            hin = st[:, :H]
            gi, gh = x @ p["Wih"].t() + p["bih"], hin @ p["Whh"].t() + p["bhh"]
            if cell == "gru":
                rr = torch.sigmoid(gi[:, :H] + gh[:, :H])
                zz = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                nn_ = torch.tanh(gi[:, 2 * H:] + rr * gh[:, 2 * H:])
                hs[k] = (1 - zz) * nn_ + zz * hin
                x = hs[k]
            else:
                a = gi + gh
                ig, fg, gg, og = torch.sigmoid(a[:, :H]), torch.sigmoid(a[:, H:2 * H]), torch.tanh(a[:, 2 * H:3 * H]), torch.sigmoid(a[:, 3 * H:])
                cn = fg * st[:, H:] + ig * gg
                x = og * torch.tanh(cn)
                hs[k] = torch.cat([x, cn], -1)
        ys.append(torch.nn.functional.layer_norm(x, (H,), g, be, 1e-5))
    y_ref = torch.cat(ys)
    (y_ref * dy).sum().backward()

    gw = gn.GruWorkspace(net, L, N, True)
    dev = lambda t: t.detach().float().to(DEV).contiguous()
    fd, md = dev(feats), dev(masks)
    y = gn.gru_forward(net, gw, fd, dev(h0), md, L, N, True)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(gw.h_last[:N].cpu().numpy(), torch.stack(hs, 1).detach().numpy(), rtol=1e-4, atol=1e-5)
    net.grad.zero_()
    dfeat = gn.gru_backward(net, gw, fd, md, dev(dy), L, N)
    sB = np.sqrt(L * N)
    np.testing.assert_allclose(dfeat.cpu().numpy(), feats.grad.numpy(), rtol=2e-4, atol=2e-4)
    checks = [(r["g"], g, (H,)), (r["be"], be, (H,))]
    for ly, p in zip(r["layers"], P):
        checks += [(ly["Wih"], p["Wih"], (G * H, H)), (ly["Whh"], p["Whh"], (G * H, H)), (ly["bih"], p["bih"], (G * H,)),
                   (ly["bhh"], p["bhh"], (G * H,))]
    for off, ref, sh in checks:
        np.testing.assert_allclose(net.v(off, *sh, grad=True).cpu().numpy(), ref.grad.numpy(), rtol=3e-4, atol=4e-5 * sB)


# ---- end to end through make / PPONet / PPOAgent ----------------------------------------------------------------------
class MatchTargetEnv:
    """Host VecEnv (the duck type of examples/isaac/isaac2openrl.py:28-88) with a MultiDiscrete([3, 2]) action space:
    the observation one-hot-encodes a target per component, the reward counts the matching components."""

    def __init__(self, n, seed=0):
        from openrl_amd import spaces

        self.n, self.rs = n, np.random.RandomState(seed)
        self.observation_space = spaces.Box(-np.inf, np.inf, (5,))
        self.action_space = spaces.MultiDiscrete([3, 2])
        self.dtypes = set()

    parallel_env_num = property(lambda s: s.n)
    agent_num = 1
    env_name = "match_target"
    use_monitor = False

    def _obs(self):
        self.t0, self.t1 = self.rs.randint(0, 3, self.n), self.rs.randint(0, 2, self.n)
        o = np.zeros((self.n, 1, 5), np.float32)
        o[np.arange(self.n), 0, self.t0] = 1.0
        o[np.arange(self.n), 0, 3 + self.t1] = 1.0
        return o

    def reset(self, seed=None, options=None):
        return self._obs(), [{} for _ in range(self.n)]

    def step(self, actions, extra_data=None):
        assert actions.shape == (self.n, 1, 2)
        self.dtypes.add(actions.dtype.kind)
        r = ((actions[:, 0, 0] == self.t0).astype(np.float32) + (actions[:, 0, 1] == self.t1)).reshape(self.n, 1, 1)
        self.last_mean_reward = float(r.mean())
        return self._obs(), r, np.zeros((self.n, 1), bool), [{} for _ in range(self.n)]

    def batch_rewards(self, buffer):
        return {}

    def close(self):
        pass


def test_multidiscrete_policy_learns_end_to_end_on_a_host_env():
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent

    cfg = default_cfg(["--episode_length", "16", "--ppo_epoch", "5", "--lr", "3e-3", "--critic_lr", "3e-3", "--seed", "3",
                       "--gamma", "0.0", "--log_interval", "100000"])
    env = MatchTargetEnv(256, seed=3)
    net = PPONet(env, cfg=cfg, device=DEV)
    assert net.module.generic and net.module.act_width == 2
    agent = PPOAgent(net)
    agent.train(total_time_steps=256 * 16 * 40)
    assert env.dtypes == {"i"}, "MultiDiscrete actions reach a host env as integers"
    assert env.last_mean_reward > 1.6, env.last_mean_reward  # 2.0 = both components always right; random = 0.83


@pytest.mark.parametrize("argv", [["--hidden_size", "128", "--layer_N", "2", "--activation_id", "0",
                                   "--use_feature_normalization", "true"],
                                  ["--use_share_model", "true"]])
def test_general_towers_train_cartpole_end_to_end_and_checkpoint(argv, tmp_path):
    """make -> PPONet -> PPOAgent.train on the device-resident CartPole with a non-default tower / the shared model
    (the fused general rollout, orl_gen_rollout_fused), then save -> load -> identical greedy actions."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent

    cfg = default_cfg(argv + ["--episode_length", "32", "--seed", "1", "--log_interval", "100000"])
    env = make("CartPole-v1", env_num=256, device=DEV, seed=1)
    net = PPONet(env, cfg=cfg, device=DEV)
    assert net.module.generic
    agent = PPOAgent(net)
    agent.train(total_time_steps=256 * 32 * 25)
    assert agent.driver.fused and agent.driver.fused_generic
    stats = env.statistics(agent.driver.buffer) if hasattr(env, "statistics") else {}
    obs = torch.randn(64, 4, device=DEV) * 0.05
    a1, _ = net.module.act(obs, None, None, deterministic=True)
    agent.save(tmp_path / "ckpt")
    for m in net.module.models.values():
        m.theta.zero_()
    agent.load(tmp_path / "ckpt")
    a2, _ = net.module.act(obs, None, None, deterministic=True)
    assert torch.equal(a1, a2)
    for m in net.module.models.values():
        assert torch.isfinite(m.theta).all()
    # it learns: mean episode length of the last rollouts is well above the ~22 steps of a random policy
    ep = [v for k, v in stats.items() if "episode" in k.lower() and "len" in k.lower()]
    if ep:
        assert float(np.mean(ep)) > 60, stats


@pytest.mark.parametrize("kind", ["mlp", "gru"])
def test_two_stream_and_one_stream_routes_give_identical_weights(kind):
    """The critic's chain on a second stream (updates of separate networks, rollout steps of recurrent towers) against
    everything on one stream: every kernel writes its own buffers in a fixed order, so the parameters after three
    iterations are equal bit for bit."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent

    def run(two_stream):
        if kind == "mlp":
            argv, name, n, T = ["--hidden_size", "128", "--layer_N", "2"], "CartPole-v1", 256, 32
        else:
            argv, name, n, T = ["--hidden_size", "128", "--use_recurrent_policy", "true", "--data_chunk_length", "5"], "simple_spread", 64, 25
        cfg = default_cfg(argv + ["--episode_length", str(T), "--seed", "3", "--log_interval", "100000", "--ppo_epoch", "3"])
        env = make(name, env_num=n, device=DEV, seed=3)
        net = PPONet(env, cfg=cfg, device=DEV)
        assert net.module.generic
        net.module.two_stream = two_stream
        agent = PPOAgent(net)
        agent.train(total_time_steps=n * T * 3)
        torch.cuda.synchronize()
        return {k: m.theta.clone() for k, m in net.module.models.items()}

    a, b = run(True), run(False)
    for k in a:
        assert torch.isfinite(a[k]).all()
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("tag", ["default", "general", "shared"])
def test_loading_a_state_dict_saved_by_the_reference_reproduces_its_outputs(tag):
    """Checkpoint compatibility in the direction that matters for a drop-in: ``state_dict()`` tensors produced by the
    REFERENCE's networks (tests/golden/state_dicts.npz) are loaded by key into the engine's networks (fused default
    tower, general tower, shared model) and the deterministic ``get_actions`` probe must come out as the reference's."""
    from collections import OrderedDict

    from openrl_amd import spaces
    from openrl_amd.modules.ppo_module import PPOModule

    g = H.load_golden("state_dicts")
    cfg = H.case_cfg({"argv": g[tag + "/argv"]})
    D = g[tag + "/probe_obs"].shape[1]
    cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = 4, 1, cfg.hidden_size
    act = {"default": spaces.Discrete(2), "general": spaces.Box(-1, 1, (3,)), "shared": spaces.Discrete(3)}[tag]
    obs_space = spaces.Box(-np.inf, np.inf, (D,))
    module = PPOModule(cfg, obs_space, obs_space, act, share_model=cfg.use_share_model, device=DEV, rank=0, world_size=1)
    assert module.generic == (tag != "default")
    for name, model in module.models.items():
        pre = "%s/%s/" % (tag, name)
        model.load_state_dict(OrderedDict((k[len(pre):], torch.tensor(v)) for k, v in g.items() if k.startswith(pre)))
    obs = g[tag + "/probe_obs"]
    v, a, lp, _, _ = module.get_actions(obs, obs, None, None, None, deterministic=True)
    np.testing.assert_allclose(v.cpu().numpy(), g[tag + "/probe_values"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(a.cpu().numpy(), g[tag + "/probe_actions"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(lp.cpu().numpy(), g[tag + "/probe_logp"], rtol=2e-4, atol=2e-5)
    vn = module.get_critic_value_normalizer()
    cname = "model" if tag == "shared" else "critic"
    np.testing.assert_allclose(vn.state.cpu().numpy(),
                               [float(g["%s/%s/value_normalizer.%s" % (tag, cname, k)].reshape(-1)[0])
                                for k in ("running_mean", "running_mean_sq", "debiasing_term")], rtol=1e-6)


@pytest.mark.parametrize("algo", ["a2c", "mat"])
def test_algorithm_variants_run_on_general_towers(algo):
    """A2CAlgorithm (policy-gradient loss bit of the hyper-parameters) and MATAlgorithm ((step, env)-pair minibatches) on
    NON-default towers: the general path's loss kernels / index handling take both; A2C's update must equal a PPO update
    with an infinite clip range in its first epoch only up to the loss definition, so the check is end-to-end sanity -
    finite losses, parameters that move, `ratio` absent for A2C."""
    from openrl_amd.configs.config import create_config_parser
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import A2CAgent, MATAgent

    cfg = create_config_parser().parse_args("--hidden_size 96 --layer_N 2 --activation_id 0 --episode_length 20 --ppo_epoch 2 "
                                            "--num_mini_batch 2".split())
    env = make("simple_spread", env_num=8)
    agent = (A2CAgent if algo == "a2c" else MATAgent)(Net(env, cfg=cfg))
    mod = agent.net.module
    assert mod.generic
    th0 = mod.models["policy"].theta.clone()
    agent.train(total_time_steps=8 * 20 * 3)
    tr = agent.driver.trainer
    assert tr.__class__.__name__ == ("A2CAlgorithm" if algo == "a2c" else "MATAlgorithm") and tr.generic
    assert not torch.equal(th0, mod.models["policy"].theta) and torch.isfinite(mod.models["policy"].theta).all()
    assert torch.isfinite(mod.models["critic"].theta).all()
    if algo == "mat":
        idx = tr.last_indices[-1].cpu().numpy().reshape(-1, 3)
        assert np.all(idx % 3 == np.arange(3)) and np.all(idx // 3 == idx[:, :1] // 3)
    env.close()


def test_model_dict_with_the_stock_network_classes_builds_the_usual_towers():
    """``PPONet(env, cfg, model_dict={"policy": PolicyNetwork, "critic": ValueNetwork})`` (ppo_net.py:57-58): the stock
    classes select the engine's built towers - same initial weights as without model_dict; a custom class is refused."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.modules.networks import PolicyNetwork, ValueNetwork
    from openrl_amd.utils.util import set_seed

    thetas = []
    for md in (None, {"policy": PolicyNetwork, "critic": ValueNetwork}):
        cfg = default_cfg(["--seed", "3"])
        env = make("CartPole-v1", env_num=4, device=DEV)
        set_seed(3)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=4, model_dict=md)
        thetas.append(net.module.models["policy"].theta.clone())
    assert torch.equal(thetas[0], thetas[1])

    class Custom(torch.nn.Module):
        pass

    with pytest.raises(NotImplementedError):
        PPONet(make("CartPole-v1", env_num=4, device=DEV), cfg=default_cfg([]), device=DEV, model_dict={"policy": Custom})


@pytest.mark.parametrize("argv,env_id,kw,N,T", [
    (["--hidden_size", "128"], "SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=7), 50, 23),
    (["--hidden_size", "128", "--layer_N", "2", "--activation_id", "0", "--use_feature_normalization", "true"],
     "SyntheticFixedStep-v0", dict(obs_dim=17, episode_limit=9, action_space="box6"), 70, 21),
    (["--hidden_size", "32", "--activation_id", "3"], "CartPole-v1", {}, 50, 23),
    (["--use_share_model", "true", "--hidden_size", "48"], "SyntheticFixedStep-v0", dict(obs_dim=5, episode_limit=6), 40, 11),
    (["--hidden_size", "256"], "SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=200), 1024, 32),
])
def test_fused_general_rollout_equals_stepwise_rollout(argv, env_id, kw, N, T):
    """``orl_gen_rollout_fused`` (all steps of {general policy tower, sampling, env.step, insert} in ONE launch + ONE
    batched critic forward over the T + 1 slots) against the stepwise route (``orl_gen_act`` + ``orl_env_step`` +
    ``orl_buffer_insert`` per step): same Philox counters and env streams, the same tile arithmetic compiled in another
    translation unit (a few ulp on float fields; a sampled class may flip only on a CDF edge)."""
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.utils.util import set_seed

    box = None
    if isinstance(kw.get("action_space"), str):
        box = int(kw["action_space"][3:])
        kw = dict(kw, action_space=spaces.Box(-1.0, 1.0, (box,)))
    bufs = []
    for mode in ("fused", "stepwise"):
        cfg = default_cfg(["--seed", "3", "--episode_length", str(T), "--amd_rollout_mode", mode] + argv)
        env = make(env_id, env_num=N, device=DEV, seed=3, **kw)
        set_seed(3)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        assert net.module.generic

        class _Agent:
            num_time_steps = 0

        cfg.num_env_steps = N * T
        trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
        buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
        agent = _Agent()
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
        assert drv.fused == (mode == "fused")
        drv.reset_and_buffer_init()
        drv.actor_rollout()
        drv.compute_returns()
        assert agent.num_time_steps == N * T and env.global_step == T and net.module.rng_step == T
        bufs.append(buf.data)
    a, b = bufs
    act_a, act_b = a.actions.cpu().numpy(), b.actions.cpu().numpy()
    if box is not None:
        np.testing.assert_allclose(act_a, act_b, rtol=1e-5, atol=2e-6)
        same = np.ones_like(act_a, dtype=bool)
    else:
        same = act_a == act_b
    assert same.mean() >= 0.999, same.mean()
    if env_id.startswith("Synthetic"):
        for f in ("policy_obs", "rewards", "masks", "active_masks", "bad_masks"):
            assert np.array_equal(getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy()), f
        np.testing.assert_allclose(a.value_preds.cpu().numpy(), b.value_preds.cpu().numpy(), rtol=2e-5, atol=2e-6)
        lp_a, lp_b = a.action_log_probs.cpu().numpy(), b.action_log_probs.cpu().numpy()
        np.testing.assert_allclose(lp_a[same], lp_b[same], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(a.returns.cpu().numpy(), b.returns.cpu().numpy(), rtol=1e-4, atol=1e-5)
    else:
        for f in ("policy_obs", "value_preds", "action_log_probs", "rewards", "masks"):
            x, y = getattr(a, f).cpu().numpy()[:3], getattr(b, f).cpu().numpy()[:3]
            np.testing.assert_allclose(x, y, rtol=2e-5, atol=2e-6, err_msg=f)

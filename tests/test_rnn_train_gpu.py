"""Recurrent (GRU) path through the host mirror of the reference API: PPOModule / ReplayData / PPOAlgorithm /
OnPolicyDriver with ``use_recurrent_policy`` (the way examples/mpe/mpe_ppo.yaml runs cfg4), replaying the golden
cases minted from the real reference.  Needs a MI355X."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import rnn_oracle as ro
from tests import helpers as H
from tests import rnn_helpers as RH

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
THETA_RTOL, THETA_ATOL = 2e-3, 4e-5

INFO_RTOL, INFO_ATOL = 3e-4, 3e-5
KEYS = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")


def _assert_thetas(g, module):
    """The bar on the UPDATE d_theta = theta_1 - theta_0 first (tests/helpers.py: cosine >= 0.9999 and 2 % of |d_ref| on
    >= 99 % of the entries), then the historical check on theta_1."""
    for name, k0, k1 in (("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")):
        got = module.models[name].theta.cpu().numpy()
        # The per-block bar needs the tower's flat layout.  Only GENERAL towers (non-default MLPBase shapes: another flat
        # layout, kept by their own module) may go without it; for the default recurrent tower a missing or mis-sized block
        # table is a test failure, not a silent fallback to the global bar.
        model = module.models[name]
        if getattr(module, "generic", False):  # PPOModule.generic: the layer-wise general path (modules/generic_net.py)
            blocks = None
        else:
            blocks = H.blocks_of(model, recurrent=True)
            assert sum(n for _, n in blocks.values()) == g[k0].size, (
                f"{name}: the block table covers {sum(n for _, n in blocks.values())} of {g[k0].size} parameters")
        H.assert_update_parity(g[k0], got, g[k1], name, blocks=blocks)  # + the per-block bar (W1 .. Wih / Whh .. b3)
        np.testing.assert_allclose(got, g[k1], rtol=THETA_RTOL, atol=THETA_ATOL)


def _spaces(g):
    from openrl_amd import spaces

    Dp, Dc = g["buf_policy_obs"].shape[-1], g["buf_critic_obs"].shape[-1]
    box = lambda d: spaces.Box(-np.inf, np.inf, (d,))
    obs_space = box(Dp) if Dp == Dc else spaces.Dict({"policy": box(Dp), "critic": box(Dc)})
    if "buf_action_masks" in g:
        act_space = spaces.Discrete(g["buf_action_masks"].shape[-1])
    else:
        act_space = spaces.Box(-1, 1, (g["buf_actions"].shape[-1],))
    return obs_space, act_space


def build_engine(g, seed=None, gemm=None):
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule

    cfg = H.case_cfg(g)
    T, N, A = g["buf_actions"].shape[:3]
    cfg.episode_length, cfg.n_rollout_threads, cfg.num_agents = T, N, A
    cfg.rnn_hidden_size = cfg.hidden_size * (2 if cfg.rnn_type == "lstm" else 1)  # modules/common/ppo_net.py:72-81
    if gemm is not None:  # the recurrent row kernel's GEMM path (cfg.amd_rnn_gemm: fp32 = default | split | split_w4)
        cfg.amd_rnn_gemm = gemm
    obs_space, act_space = _spaces(g)
    if seed is not None:
        import random
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=DEV, rank=0, world_size=1)
    buf = ReplayData(cfg, A, obs_space, act_space, device=DEV)
    algo = PPOAlgorithm(cfg, module, agent_num=A, device=DEV)
    return cfg, module, buf, algo


@pytest.mark.parametrize("case,seed", [("train_recurrent", 5), ("train_recurrent_chunk5", 6)])
def test_recurrent_module_init_is_the_reference_init(case, seed):
    g = H.load_golden(case)
    _, module, _, _ = build_engine(g, seed=seed)
    # same generator stream; the orthogonal init's QR runs in this host's LAPACK, hence allclose and not array_equal
    # (bit-equality on the authoring host is asserted by tests/test_rnn_oracle_cpu.py for the same code path)
    np.testing.assert_allclose(module.models["policy"].theta.cpu().numpy(), g["theta_p0"], rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(module.models["critic"].theta.cpu().numpy(), g["theta_c0"], rtol=1e-4, atol=3e-5)
    sd = module.models["policy"].state_dict()
    assert sd["rnn.rnn.weight_ih_l0"].shape == (192, 64) and sd["rnn.norm.bias"].shape == (64,)


@pytest.mark.parametrize("gemm", [H.experimental("split"), "fp32", "fp32_recompute", H.experimental("split_w4")])
@pytest.mark.parametrize("case", RH.RNN_CASES)
def test_recurrent_train_matches_reference_golden(case, gemm):
    """gemm: the row kernel (cfg.amd_rnn_gemm) - fp32 (default: the register-resident kernel of csrc/orl_rnn_l2.h for chunks of
    2 steps, the recompute kernel for other lengths), fp32_recompute (the recompute kernel whatever the length) and the
    streamed bf16-split build with 8 or (split_w4) 4 waves per workgroup; all must land on the reference's update."""
    g = H.load_golden(case)
    cfg, module, buf, algo = build_engine(g, gemm=gemm)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
              "bad_masks", "active_masks", "action_masks", "rnn_states", "rnn_states_critic"):
        if "buf_" + f in g and getattr(buf, f) is not None:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    r = RH.rnn_oracle_replay(g)
    assert len(algo.last_indices) == len(r["used"])
    for got, want in zip(algo.last_indices, r["used"]):  # torch.randperm(data_chunks) stream of replay_data.py:1078
        assert np.array_equal(got.cpu().numpy(), want)
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
    _assert_thetas(g, module)
    np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)


@pytest.mark.parametrize("case", ["train_recurrent", "train_recurrent_chunk5"])
def test_update_parity_bar_rejects_a_short_recurrent_update(case):
    """Negative control of the d_theta bar on the recurrent engine: the golden case with its last epoch skipped, and the
    full update with the GRU's input-weight block left at theta_0, must be refused."""
    g = H.load_golden(case)

    def run(skip):
        cfg, module, buf, algo = build_engine(g)
        algo.ppo_epoch -= skip
        module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
        module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
        for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
                  "bad_masks", "active_masks", "action_masks", "rnn_states", "rnn_states_critic"):
            if "buf_" + f in g and getattr(buf, f) is not None:
                getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
        torch.manual_seed(int(g["perm_seed"]))
        algo.prep_training()
        algo.train(buf)
        return module

    full = run(0)
    _assert_thetas(g, full)
    sd = full.models["policy"].state_dict()
    flat = full.models["policy"].theta.cpu().numpy().copy()
    # the GRU's W_ih block (192 x 64) inside the flat vector: found by value, reset to theta_0
    wih = sd["rnn.rnn.weight_ih_l0"].detach().cpu().numpy().ravel()
    pos = next(o for o in range(0, flat.size - wih.size + 1) if flat[o] == wih[0] and np.array_equal(flat[o:o + wih.size], wih))
    broken = flat.copy()
    broken[pos:pos + wih.size] = g["theta_p0"][pos:pos + wih.size]
    H.assert_update_parity_rejects(g["theta_p0"], broken, g["theta_p1"], "recurrent policy without dW_ih")
    short = run(1)
    for name, k0, k1 in (("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")):
        H.assert_update_parity_rejects(g[k0], short.models[name].theta.cpu().numpy(), g[k1], name + ", last epoch skipped")


@pytest.mark.parametrize("case", ["train_recurrent", "train_recurrent_chunk5"])
def test_recurrent_turn_on_false_updates_only_the_critic(case):
    """turn_on=False through the recurrent update: the policy tower (incl. its LayerNorm-affine dot workgroups) gets no
    gradient and no Adam step, the critic gets exactly its usual update."""
    g = H.load_golden(case)
    res = []
    for turn_on in (True, False):
        cfg, module, buf, algo = build_engine(g)
        module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
        module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
        for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
                  "bad_masks", "active_masks", "action_masks", "rnn_states", "rnn_states_critic"):
            if "buf_" + f in g and getattr(buf, f) is not None:
                getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
        algo.perm_mode = "device"
        info = algo.train(buf, turn_on=turn_on)
        res.append((module.models["policy"].theta.clone(), module.models["critic"].theta.clone(), dict(info),
                    module.models["policy"].grad.clone()))
    on, off = res
    assert torch.equal(off[0], torch.tensor(g["theta_p0"], device=DEV)) and not torch.equal(on[0], off[0])
    assert torch.equal(on[1], off[1]) and torch.all(off[3] == 0)
    assert off[2]["actor_grad_norm"] == 0.0 and off[2]["value_loss"] == on[2]["value_loss"]


def test_recurrent_stepwise_rollout_reproduces_the_reference_buffer():
    """Teacher-forced stepwise rollout (golden actions are deterministic functions of obs, states and the sampler's
    uniforms, which the reference drew from torch.multinomial - so compare what does not depend on the sample:
    hidden states and values slot by slot, given the golden observations / masks)."""
    g = H.load_golden("train_recurrent")
    cfg, module, buf, algo = build_engine(g)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    T, N, A = g["buf_actions"].shape[:3]
    for f in ("policy_obs", "critic_obs", "masks"):
        getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    for t in range(T):
        v, a, lp, hp, hc = module.get_actions(buf.get_batch_data("critic_obs", t), buf.get_batch_data("policy_obs", t),
                                              buf.get_batch_data("rnn_states", t), buf.get_batch_data("rnn_states_critic", t),
                                              buf.get_batch_data("masks", t))
        assert hp.shape == (N * A, 1, 64) and a.shape == (N * A, 1)
        m = buf.masks[t + 1].unsqueeze(-1)
        buf.rnn_states[t + 1].copy_(hp.view(N, A, 1, 64) * m)      # driver: zero where the env finished
        buf.rnn_states_critic[t + 1].copy_(hc.view(N, A, 1, 64) * m)
        np.testing.assert_allclose(v.cpu().numpy().reshape(N, A, 1), g["buf_value_preds"][t], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states.cpu().numpy(), g["buf_rnn_states"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states_critic.cpu().numpy(), g["buf_rnn_states_critic"], rtol=3e-4, atol=3e-5)
    nv = module.get_values(buf.get_batch_data("critic_obs", -1), buf.rnn_states_critic[-1].reshape(-1, 64),
                           buf.masks[-1].reshape(-1, 1))
    np.testing.assert_allclose(nv.cpu().numpy().reshape(N, A, 1), g["next_values"], rtol=3e-4, atol=3e-5)


def test_recurrent_agent_trains_on_a_host_multiagent_env():
    """make-less end to end: PPONet + PPOAgent.train with use_recurrent_policy on the duck-typed host env."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent
    from tests.test_multiagent_gpu import ToyMultiAgentEnv

    N, A, T = 8, 3, 25
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "2", "--use_recurrent_policy", "true", "--seed", "1"])
    env = ToyMultiAgentEnv(N, A)
    net = PPONet(env, cfg=cfg, device=DEV)
    agent = PPOAgent(net)
    th0 = net.module.models["policy"].theta.clone()
    agent.train(total_time_steps=2 * N * T)
    d = agent.driver.buffer.data
    assert d.rnn_states.shape == (T + 1, N, A, 1, 64) and d.rnn_states.abs().max() > 0
    assert torch.all(d.rnn_states[1:, 0] == 0)          # env 0 finishes every step -> its stored states are zero
    assert torch.isfinite(net.module.models["policy"].theta).all()
    assert (net.module.models["policy"].theta - th0).abs().max() > 0
    obs = {"policy": np.zeros((N, A, 18), np.float32), "critic": np.zeros((N, A, 54), np.float32)}
    agent.reset()
    action, _ = agent.act(obs)
    assert action.shape == (N, A, 1)


def _fill(buf, g):
    for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
              "bad_masks", "active_masks", "action_masks", "rnn_states", "rnn_states_critic"):
        if "buf_" + f in g and getattr(buf, f) is not None:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))


@pytest.mark.parametrize("perm_mode", ["reference", "device"])
@pytest.mark.parametrize("case", ["train_recurrent_jrpo", "train_recurrent_gen_jrpo"])
def test_jrpo_train_matches_reference_golden(case, perm_mode):
    """use_joint_action_loss (JRPO): recurrent_generator_v3 + joint ratio over agents, critic on agent 0 only
    (algorithms/ppo.py:254-300, buffers/replay_data.py:425-551) - the engine's pre-pass + adjusted-records update against
    the REAL reference's PPOAlgorithm.train: the default recurrent tower (fused kernels) and a general one (hidden 96,
    layer_N 2: layer-wise update with the GRU, the critic on agent 0's rows)."""
    g = H.load_golden(case)
    cfg, module, buf, algo = build_engine(g)
    assert algo.use_joint_action_loss and algo.recurrent and module.generic == ("gen" in case)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    _fill(buf, g)
    algo.perm_mode = perm_mode
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    T, N, A = g["buf_actions"].shape[:3]
    n_chunks = (N * T // cfg.data_chunk_length) // cfg.num_mini_batch
    assert len(algo.last_indices) == cfg.ppo_epoch * cfg.num_mini_batch and algo.last_indices[0].numel() == n_chunks
    if perm_mode == "reference":  # with two minibatches per epoch only the reference's own chunk order is comparable
        np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
        _assert_thetas(g, module)
        np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)
    else:
        assert np.isfinite([info[k] for k in KEYS]).all()
        assert abs(info["ratio"] - 1.0) < 0.05 and 0.5 < info["dist_entropy"] < 1.7


def test_recurrent_evaluate_actions_vs_oracle():
    """PPOModule.evaluate_actions with recurrent networks outside the update (ppo_module.py:149-193, rnn.py:39-99):
    L steps of N sequences flattened [L*N, ...], states [N, H] entering step 0 - values, log-probs and the masked mean
    entropy against the oracle's unrolled GRU towers on a golden chunk sample."""
    g = H.load_golden("train_recurrent")
    cfg, module, buf, algo = build_engine(g)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    pspec, cspec = RH.rnn_specs(g)
    b = H.case_buffer(g)
    adv = np.zeros_like(b["rewards"])
    rows = ro.buffer_rows(b, adv)
    L = 2
    chunks = np.arange(rows["adv"].shape[0] // L)[::3]
    s = ro.chunk_sample(rows, chunks, L)
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    pth, cth = torch.tensor(g["theta_p0"]), torch.tensor(g["theta_c0"])
    want_v, _ = ro.rnn_tower_forward(cspec, cth, t(s["critic_obs"]), t(s["rnn_states_critic"]), t(s["masks"]))
    out, _ = ro.rnn_tower_forward(pspec, pth, t(s["policy_obs"]), t(s["rnn_states"]), t(s["masks"]))
    dist = torch.distributions.Categorical(logits=po.masked_logits(out, t(s["action_masks"])))
    want_lp = dist.log_prob(t(s["actions"]).squeeze(-1).long()).unsqueeze(-1)
    am = t(s["active_masks"])
    want_ent = (dist.entropy() * am.squeeze(-1)).sum() / am.sum()
    v, lp, ent, _ = module.evaluate_actions(s["critic_obs"], s["policy_obs"], s["rnn_states"], s["rnn_states_critic"],
                                            s["actions"], s["masks"], s["action_masks"], s["active_masks"])
    np.testing.assert_allclose(v.cpu().numpy(), want_v.detach().numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(lp.cpu().numpy(), want_lp.detach().numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(float(ent), float(want_ent), rtol=1e-4)


# ---- recurrent GENERAL towers (hidden_size / layer_N / activation / feature norm + the GRU): modules/generic_net.py ----
GEN_RNN_CASES = [("train_recurrent_gen_h128", 31), ("train_recurrent_gen_l2_tanh_fn", 32), ("train_recurrent_gen_n2", 34),
                 ("train_recurrent_gen_lstm", 35), ("train_recurrent_gen_lstm_n2", 36)]
_BUF_FIELDS = ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
               "bad_masks", "active_masks", "action_masks", "rnn_states", "rnn_states_critic")


def _load_case(g, buf, module):
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    for f in _BUF_FIELDS:
        if "buf_" + f in g and getattr(buf, f) is not None:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))


@pytest.mark.parametrize("case,seed", GEN_RNN_CASES)
def test_general_recurrent_towers_init_and_state_dict_are_the_reference_s(case, seed):
    g = H.load_golden(case)
    cfg, module, _, _ = build_engine(g, seed=seed)
    assert module.generic and module.recurrent
    np.testing.assert_allclose(module.models["policy"].theta.cpu().numpy(), g["theta_p0"], rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(module.models["critic"].theta.cpu().numpy(), g["theta_c0"], rtol=1e-4, atol=3e-5)
    Hs = cfg.hidden_size
    sd = module.models["critic"].state_dict()
    G = 4 if cfg.rnn_type == "lstm" else 3
    assert sd["rnn.rnn.weight_hh_l0"].shape == (G * Hs, Hs) and sd["rnn.norm.weight"].shape == (Hs,)
    assert ("rnn.rnn.bias_hh_l1" in sd) == (cfg.recurrent_N == 2)
    keys = [k for k in sd if not k.startswith("value_normalizer")]
    assert keys.index("rnn.rnn.weight_ih_l0") < keys.index("v_out.weight") and keys[-1] == "v_out.bias"


@pytest.mark.parametrize("case,seed", GEN_RNN_CASES)
def test_general_recurrent_train_matches_reference_golden(case, seed):
    """PPOAlgorithm.train on recurrent general towers (trunk layer kernels + GRU BPTT over the chunks) vs the reference's
    own train(): the chunk permutation stream, the final weights of both towers, train_info and the ValueNorm state."""
    g = H.load_golden(case)
    cfg, module, buf, algo = build_engine(g)
    _load_case(g, buf, module)
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    T, N, A = g["buf_actions"].shape[:3]
    chunks = T * N * A // cfg.data_chunk_length
    torch.manual_seed(int(g["perm_seed"]))
    mbs = chunks // cfg.num_mini_batch
    want = []
    for _ in range(cfg.ppo_epoch):  # replay_data.py:1078-1082
        perm = torch.randperm(chunks).numpy()
        want += [perm[i * mbs:(i + 1) * mbs] for i in range(cfg.num_mini_batch)]
    assert len(algo.last_indices) == len(want)
    for got, w in zip(algo.last_indices, want):
        assert np.array_equal(got.cpu().numpy(), w)
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
    _assert_thetas(g, module)
    np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)


@pytest.mark.parametrize("case,seed", GEN_RNN_CASES)
def test_general_recurrent_stepwise_rollout_reproduces_the_reference_buffer(case, seed):
    """Teacher-forced get_actions step by step: hidden states and values of every slot against the reference's buffer."""
    g = H.load_golden(case)
    cfg, module, buf, algo = build_engine(g)
    _load_case(g, buf, module)
    buf.rnn_states.zero_(); buf.rnn_states_critic.zero_()
    T, N, A = g["buf_actions"].shape[:3]
    Hs, rN = cfg.rnn_hidden_size, cfg.recurrent_N   # state width: H, or 2 H = [h | c] for an LSTM
    for t in range(T):
        v, a, lp, hp, hc = module.get_actions(buf.get_batch_data("critic_obs", t), buf.get_batch_data("policy_obs", t),
                                              buf.get_batch_data("rnn_states", t), buf.get_batch_data("rnn_states_critic", t),
                                              buf.get_batch_data("masks", t))
        assert hp.shape == (N * A, rN, Hs)
        m = buf.masks[t + 1].unsqueeze(-1)
        buf.rnn_states[t + 1].copy_(hp.view(N, A, rN, Hs) * m)
        buf.rnn_states_critic[t + 1].copy_(hc.view(N, A, rN, Hs) * m)
        np.testing.assert_allclose(v.cpu().numpy().reshape(N, A, 1), g["buf_value_preds"][t], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states.cpu().numpy(), g["buf_rnn_states"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states_critic.cpu().numpy(), g["buf_rnn_states_critic"], rtol=3e-4, atol=3e-5)
    nv = module.get_values(buf.get_batch_data("critic_obs", -1), buf.rnn_states_critic[-1].reshape(-1, Hs),
                           buf.masks[-1].reshape(-1, 1))
    np.testing.assert_allclose(nv.cpu().numpy().reshape(N, A, 1), g["next_values"], rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("argv", [["--hidden_size", "128"], ["--hidden_size", "32", "--layer_N", "2", "--activation_id", "0",
                                                             "--use_naive_recurrent_policy", "true", "--use_recurrent_policy", "false"],
                                  ["--recurrent_N", "2"], ["--rnn_type", "lstm", "--hidden_size", "96"]])
def test_general_recurrent_agent_trains_mpe_end_to_end(argv, tmp_path):
    """make / PPONet / PPOAgent.train with a recurrent policy on NON-default towers on the device MPE env: the stepwise
    rollout (captured into a hipGraph after the first iteration) carries the GRU states through the buffer, the update
    runs the chunked BPTT; save / load round trip."""
    from openrl_amd.configs.config import create_config_parser
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

    base = ["--episode_length", "25", "--ppo_epoch", "2", "--num_mini_batch", "2", "--use_recurrent_policy", "true"]
    cfg = create_config_parser().parse_args(base + argv)
    env = make("simple_spread", env_num=16)
    agent = Agent(Net(env, cfg=cfg))
    mod = agent.net.module
    assert mod.generic and mod.recurrent
    th0 = mod.models["policy"].theta.clone()
    agent.train(total_time_steps=16 * 25 * 3)
    assert agent.driver._graph is not None and not agent.driver.fused
    assert not torch.equal(th0, mod.models["policy"].theta) and torch.isfinite(mod.models["policy"].theta).all()
    assert float(agent.driver.buffer.data.rnn_states.abs().sum()) > 0
    agent.save(str(tmp_path / "a"))
    agent.load(str(tmp_path / "a"))
    obs, info = env.reset(seed=1)
    action, _ = agent.act(obs, deterministic=True)
    assert action.shape == (16, 3, 1)
    env.close()


def test_shared_recurrent_network_matches_reference_golden():
    """use_share_model + use_recurrent_policy: ONE PolicyValueNetwork whose GRU serves the actor pass (rnn_states) and the
    critic pass (rnn_states_critic) - policy_value_network.py:85-91,113-172.  Init, the update (two passes through the
    shared trunk + GRU, gradients summed, both clips) and the stored states of a stepwise rollout vs the reference."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule

    g = H.load_golden("train_share_recurrent")
    cfg = H.case_cfg(g)
    T, N, A = g["buf_actions"].shape[:3]
    cfg.episode_length, cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = T, N, A, cfg.hidden_size
    obs_space, act_space = _spaces(g)
    import random
    random.seed(37); np.random.seed(37); torch.manual_seed(37)
    module = PPOModule(cfg, obs_space, obs_space, act_space, share_model=True, device=DEV, rank=0, world_size=1)
    model = module.models["model"]
    assert module.generic and module.recurrent and module.share_model
    np.testing.assert_allclose(model.reference_flat().cpu().numpy(), g["theta_m0"], rtol=1e-4, atol=3e-5)
    model.load_reference_flat(g["theta_m0"])
    buf = ReplayData(cfg, A, obs_space, act_space, device=DEV)
    for f in _BUF_FIELDS:
        if "buf_" + f in g and getattr(buf, f) is not None:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    algo = PPOAlgorithm(cfg, module, agent_num=A, device=DEV)
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
    H.assert_update_parity(g["theta_m0"], model.reference_flat().cpu().numpy(), g["theta_m1"], "shared model")
    np.testing.assert_allclose(model.reference_flat().cpu().numpy(), g["theta_m1"], rtol=THETA_RTOL, atol=THETA_ATOL)
    np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)
    # teacher-forced stepwise rollout on the initial weights: both state streams of the buffer
    model.load_reference_flat(g["theta_m0"])
    buf.rnn_states.zero_(); buf.rnn_states_critic.zero_()
    Hs = cfg.hidden_size
    for t in range(T):
        v, a, lp, hp, hc = module.get_actions(buf.get_batch_data("critic_obs", t), buf.get_batch_data("policy_obs", t),
                                              buf.get_batch_data("rnn_states", t), buf.get_batch_data("rnn_states_critic", t),
                                              buf.get_batch_data("masks", t))
        m = buf.masks[t + 1].unsqueeze(-1)
        buf.rnn_states[t + 1].copy_(hp.view(N, A, 1, Hs) * m)
        buf.rnn_states_critic[t + 1].copy_(hc.view(N, A, 1, Hs) * m)
        np.testing.assert_allclose(v.cpu().numpy().reshape(N, A, 1), g["buf_value_preds"][t], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states.cpu().numpy(), g["buf_rnn_states"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(buf.rnn_states_critic.cpu().numpy(), g["buf_rnn_states_critic"], rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("perm_mode,gemm", [H.experimental("device", "split"), H.experimental("identity", "split"),
                                            H.experimental("reference", "split"),
                                            ("device", "fp32"), ("identity", "fp32"), ("reference", "fp32"),
                                            ("device", "fp32_recompute"), H.experimental("device", "split_w4")])
def test_full_size_recurrent_update_matches_reference_golden(perm_mode, gemm):
    """BASELINE.json configs[3] at FULL size: 2048 envs x 3 agents x 25 steps = 153 600 rows = 76 800 chunks of 2 (the
    odd T makes chunks straddle lanes), Dict obs 18 / 54, Discrete(5), GRU, adv-normalise on (examples/mpe/mpe_ppo.yaml),
    3 epochs through the REAL reference's ``recurrent_generator`` + ``PPOAlgorithm.train`` (oracle/gen_golden.py::
    _train_case_full_general; reference buffers/replay_data.py:1062-1258, algorithms/ppo.py:383-458).  This is the
    batch of benchmarks/cfg4_mpe_bench.py: several chunks per wave with a ragged last round."""
    from tests.test_ppo_update_gpu import _full_general_engine

    g = H.load_golden("train_cfg4_full")
    cfg, module, buf, algo = _full_general_engine(g, perm_mode, gemm)
    assert algo.recurrent and not algo.generic
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"], rtol=INFO_RTOL, atol=INFO_ATOL)
    _assert_thetas(g, module)
    np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)


@pytest.mark.parametrize("env_id,kw,N,T", [
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=7), 50, 23),
    ("SyntheticFixedStep-v0", dict(obs_dim=17, episode_limit=9, action_space="box6"), 70, 21),
    ("CartPole-v1", {}, 96, 25),
    ("SyntheticFixedStep-v0", dict(obs_dim=18, episode_limit=50, action_space="disc9"), 1000, 32),
])
def test_fused_recurrent_rollout_on_single_agent_envs_equals_stepwise(env_id, kw, N, T):
    """``orl_rnn_rollout_fused`` on the single-agent device envs (one wave per 16-env tile with the hidden state in
    registers + the critic sweep over the stored observations) against the stepwise recurrent rollout
    (``orl_rnn_act_step`` + ``orl_env_step`` + ``orl_buffer_insert`` per step): same Philox counters and env streams -
    actions (>= 99.9 %), observations / rewards / masks exact on the synthetic env; hidden states, values, log-probs to
    fp32 round-off (weights from LDS vs from L2: another summation order)."""
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.utils.util import set_seed

    box = None
    sp = kw.get("action_space")
    if isinstance(sp, str):
        k = int(sp[4:]) if sp.startswith("disc") else int(sp[3:])
        box = k if sp.startswith("box") else None
        kw = dict(kw, action_space=spaces.Box(-1.0, 1.0, (k,)) if box else spaces.Discrete(k))
    bufs = []
    for mode in ("fused", "stepwise"):
        cfg = default_cfg(["--seed", "3", "--episode_length", str(T), "--amd_rollout_mode", mode, "--use_recurrent_policy",
                           "true"])
        env = make(env_id, env_num=N, device=DEV, seed=3, **kw)
        set_seed(3)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        assert net.module.recurrent and not net.module.generic

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
        np.testing.assert_allclose(act_a, act_b, rtol=3e-5, atol=3e-6)
        same = np.ones_like(act_a, dtype=bool)
    else:
        same = act_a == act_b
    assert same.mean() >= 0.999, same.mean()
    if env_id.startswith("Synthetic"):
        for f in ("policy_obs", "rewards", "masks", "active_masks", "bad_masks"):
            assert np.array_equal(getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy()), f
        for f in ("value_preds", "rnn_states", "rnn_states_critic", "returns"):
            np.testing.assert_allclose(getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy(), rtol=3e-4, atol=3e-5,
                                       err_msg=f)
        lp_a, lp_b = a.action_log_probs.cpu().numpy(), b.action_log_probs.cpu().numpy()
        np.testing.assert_allclose(lp_a[same], lp_b[same], rtol=3e-4, atol=3e-5)
    else:
        for f in ("policy_obs", "value_preds", "action_log_probs", "rewards", "masks", "rnn_states"):
            x, y = getattr(a, f).cpu().numpy()[:3], getattr(b, f).cpu().numpy()[:3]
            np.testing.assert_allclose(x, y, rtol=3e-4, atol=3e-5, err_msg=f)

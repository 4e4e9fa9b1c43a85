"""Algorithm variants of PPO.  MATAlgorithm (openrl/algorithms/mat.py) on the MLP PPOModule: oracle and HIP engine replay
goldens minted from the reference's own MATAlgorithm.train with 2-3 agents.
A2CAlgorithm (openrl/algorithms/a2c.py): oracle and HIP engine replay the golden case minted from the reference's
own A2CAlgorithm.train (policy loss -adv*logp, num_mini_batch forced to 1, no ``ratio`` in train_info)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

KEYS = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm")


def test_a2c_oracle_replay_matches_reference():
    g = H.load_golden("train_a2c")
    r = H.oracle_replay(g)
    assert len(r["used"]) == 3  # ppo_epoch 3 x ONE minibatch although the cfg says num_mini_batch 4
    np.testing.assert_allclose(r["ptheta"], g["theta_p1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r["ctheta"], g["theta_c1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(np.array([r["info"][k] for k in KEYS]), g["train_info"][:5], rtol=1e-5, atol=1e-6)
    assert r["info"]["ratio"] == 0.0 and g["train_info"][5] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("skip_epochs", [0, 1])
def test_a2c_engine_matches_reference_golden(skip_epochs):
    """skip_epochs = 1 is the negative control of the d_theta bar: the same run one epoch short must be refused."""
    from openrl_amd import spaces
    from openrl_amd.algorithms.a2c import A2CAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule

    dev = "cuda:0"
    g = H.load_golden("train_a2c")
    cfg = H.case_cfg(g)
    T, N, D = g["buf_policy_obs"].shape[0] - 1, g["buf_policy_obs"].shape[1], g["buf_policy_obs"].shape[-1]
    cfg.episode_length, cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = T, N, 1, cfg.hidden_size
    obs_space, act_space = spaces.Box(-np.inf, np.inf, (D,)), spaces.Discrete(g["buf_action_masks"].shape[-1])
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=dev, rank=0, world_size=1)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    buf = ReplayData(cfg, 1, obs_space, act_space, device=dev)
    for f in ("policy_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "bad_masks",
              "active_masks", "action_masks"):
        getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    algo = A2CAlgorithm(cfg, module, agent_num=1, device=dev)
    assert algo.num_mini_batch == 1
    algo.ppo_epoch -= skip_epochs
    torch.manual_seed(int(g["perm_seed"]))
    info = algo.train(buf)
    if skip_epochs:
        for name, k0, k1 in (("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")):
            H.assert_update_parity_rejects(g[k0], module.models[name].theta.cpu().numpy(), g[k1],
                                           name + ", last epoch skipped")
        return
    assert "ratio" not in info and len(algo.last_indices) == 3
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"][:5], rtol=2e-4, atol=2e-5)
    for name, k0, k1 in (("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")):
        got_flat = module.models[name].theta.cpu().numpy()
        H.assert_update_parity(g[k0], got_flat, g[k1], name)  # the bar on d_theta (tests/helpers.py)
        np.testing.assert_allclose(got_flat, g[k1], rtol=2e-3, atol=3e-5)


# ---- MATAlgorithm: one summed loss (same gradients with separate towers) + the transformer generator ----------------
MAT_CASES = ["train_mat", "train_mat_box"]
ALL_KEYS = KEYS + ("ratio",)


def _mat_specs(g):
    from oracle import ppo_oracle as po

    D = g["buf_policy_obs"].shape[-1]
    if "buf_action_masks" in g:
        return po.TowerSpec(D, g["buf_action_masks"].shape[-1], po.HEAD_CATEGORICAL), po.TowerSpec(D, 1, po.HEAD_VALUE)
    return po.TowerSpec(D, g["buf_actions"].shape[-1], po.HEAD_GAUSSIAN), po.TowerSpec(D, 1, po.HEAD_VALUE)


@pytest.mark.parametrize("case", MAT_CASES)
def test_mat_oracle_replay_matches_reference(case):
    from oracle import ppo_oracle as po

    g = H.load_golden(case)
    A = int(g["agents"])
    cfg = H.case_cfg(g)
    hp = po.hyper_from_cfg(cfg)
    pspec, cspec = _mat_specs(g)
    ptheta, ctheta = torch.tensor(g["theta_p0"]).clone(), torch.tensor(g["theta_c0"]).clone()
    padam = po.AdamOracle(ptheta.numel(), cfg.lr, cfg.opti_eps, cfg.weight_decay)
    cadam = po.AdamOracle(ctheta.numel(), cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)
    vn = po.ValueNormOracle() if cfg.use_valuenorm else None
    torch.manual_seed(int(g["perm_seed"]))
    info, _, used = po.train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, H.case_buffer(g), cfg.ppo_epoch,
                                 cfg.num_mini_batch, index_fn=po.transformer_indices(A))
    assert len(used) == cfg.ppo_epoch * cfg.num_mini_batch
    for idx in used:  # whole (step, env) pairs, agents in order
        grp = idx.reshape(-1, A)
        assert np.all(grp % A == np.arange(A)) and np.all(grp // A == grp[:, :1] // A)
    np.testing.assert_allclose(ptheta.numpy(), g["theta_p1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ctheta.numpy(), g["theta_c1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(np.array([info[k] for k in ALL_KEYS]), g["train_info"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("perm_mode", ["reference", "device", "identity"])
@pytest.mark.parametrize("case", MAT_CASES)
def test_mat_engine_matches_reference_golden(case, perm_mode):
    from openrl_amd import spaces
    from openrl_amd.algorithms.mat import MATAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule
    from oracle import ppo_oracle as po

    dev = "cuda:0"
    g = H.load_golden(case)
    A = int(g["agents"])
    cfg = H.case_cfg(g)
    if perm_mode != "reference" and cfg.num_mini_batch != 1:
        pytest.skip("another permutation lands on the reference's weights only with one minibatch per epoch")
    T, N, D = g["buf_policy_obs"].shape[0] - 1, g["buf_policy_obs"].shape[1], g["buf_policy_obs"].shape[-1]
    cfg.episode_length, cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = T, N, A, cfg.hidden_size
    obs_space = spaces.Box(-np.inf, np.inf, (D,))
    act_space = (spaces.Discrete(g["buf_action_masks"].shape[-1]) if "buf_action_masks" in g
                 else spaces.Box(-1, 1, (g["buf_actions"].shape[-1],)))
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=dev, rank=0, world_size=1)
    module.models["policy"].theta.copy_(torch.tensor(g["theta_p0"]))
    module.models["critic"].theta.copy_(torch.tensor(g["theta_c0"]))
    buf = ReplayData(cfg, A, obs_space, act_space, device=dev)
    for f in ("policy_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "bad_masks",
              "active_masks", "action_masks"):
        if "buf_" + f in g:
            getattr(buf, f).copy_(torch.tensor(g["buf_" + f]))
    algo = MATAlgorithm(cfg, module, agent_num=A, device=dev)
    algo.perm_mode = perm_mode
    torch.manual_seed(int(g["perm_seed"]))
    algo.prep_training()
    info = algo.train(buf)
    if perm_mode == "reference":  # the same (step, env) pairs in the same order as the reference's generator
        torch.manual_seed(int(g["perm_seed"]))
        fn = po.transformer_indices(A)
        want = [i for _ in range(cfg.ppo_epoch) for i in fn(T * N * A, cfg.num_mini_batch)]
        assert len(algo.last_indices) == len(want)
        for got, w in zip(algo.last_indices, want):
            assert np.array_equal(got.cpu().numpy(), w)
    np.testing.assert_allclose(np.array([info[k] for k in ALL_KEYS]), g["train_info"], rtol=2e-4, atol=2e-5)
    for name, k0, k1 in (("policy", "theta_p0", "theta_p1"), ("critic", "theta_c0", "theta_c1")):
        got_flat = module.models[name].theta.cpu().numpy()
        H.assert_update_parity(g[k0], got_flat, g[k1], name)  # the bar on d_theta (tests/helpers.py)
        np.testing.assert_allclose(got_flat, g[k1], rtol=2e-3, atol=3e-5)
    if "vn_state1" in g:
        np.testing.assert_allclose(module.get_critic_value_normalizer().state.cpu().numpy(), g["vn_state1"], rtol=1e-5)

"""A2CAlgorithm (openrl/algorithms/a2c.py): oracle and HIP engine replay the golden case minted from the reference's
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
def test_a2c_engine_matches_reference_golden():
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
    torch.manual_seed(int(g["perm_seed"]))
    info = algo.train(buf)
    assert "ratio" not in info and len(algo.last_indices) == 3
    np.testing.assert_allclose(np.array([info[k] for k in KEYS]), g["train_info"][:5], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(module.models["policy"].theta.cpu().numpy(), g["theta_p1"], rtol=2e-3, atol=3e-5)
    np.testing.assert_allclose(module.models["critic"].theta.cpu().numpy(), g["theta_c1"], rtol=2e-3, atol=3e-5)

"""Generate ``tests/golden/*.npz`` by running the REAL reference (``/root/reference/openrl``) on CPU.

TEST INFRASTRUCTURE.  Runs only where ``/root/reference`` exists (the authoring container):

    python -m oracle.gen_golden

The reference's drivers/runners/envs cannot be imported offline (gymnasium, wandb, pettingzoo are
missing - SURVEY.md section 8c), so the rollout part follows ``openrl/drivers/onpolicy_driver.py:154-203``
by hand while every numeric object is the reference's own: ``PPOModule``, ``NormalReplayBuffer``
(``ReplayData.insert/compute_returns/feed_forward_generator``), ``PPOAlgorithm.train``, ``ValueNorm``.
The vectors are small (a few hundred KB in total) and committed; the GPU box never needs the reference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import ref_stubs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _flat(model) -> np.ndarray:
    # trainable parameters in registration order; ValueNetwork also registers the three ValueNorm state
    # scalars as requires_grad=False Parameters (valuenorm.py:24-35) - they are recorded as vn_state1
    return torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad]).numpy().copy()


def _cfg(argv, N, T):
    cfg = ref_stubs.reference_cfg(argv)
    cfg.num_agents = 1
    cfg.n_rollout_threads = N
    cfg.learner_n_rollout_threads = N
    cfg.rnn_hidden_size = cfg.hidden_size * (2 if cfg.rnn_type == "lstm" else 1)  # modules/common/ppo_net.py:72-81
    cfg.episode_length = T
    return cfg


def gae_cases():
    from gymnasium.spaces import Box, Discrete
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.utils.valuenorm import ValueNorm

    out = {}
    # known-answer vector of SURVEY.md section 8c
    for proper in (False, True):
        cfg = _cfg(["--use_proper_time_limits", str(proper), "--use_valuenorm", "false"], 2, 4)
        buf = NormalReplayBuffer(cfg, 1, Box(-1, 1, (4,)), Discrete(2), data_client=None).data
        buf.rewards[:] = np.array([[1, .5], [1, -1], [1, 2], [1, .25]], np.float32).reshape(4, 2, 1, 1)
        buf.value_preds[:4] = np.array([[.5, .1], [.4, -.2], [.3, .7], [.2, 0]], np.float32).reshape(4, 2, 1, 1)
        buf.masks[:] = np.array([[1, 1], [1, 1], [1, 0], [1, 1], [1, 1]], np.float32).reshape(5, 2, 1, 1)
        buf.bad_masks[:] = np.array([[1, 1], [1, 1], [1, 1], [1, 0], [1, 1]], np.float32).reshape(5, 2, 1, 1)
        buf.compute_returns(np.array([.1, .9], np.float32).reshape(2, 1, 1), None)
        out["kat_proper%d_returns" % proper] = buf.returns.copy()
    # random cases: use_gae x proper x valuenorm
    rs = np.random.RandomState(7)
    T, N, A = 9, 5, 2
    base = dict(rewards=rs.randn(T, N, A, 1).astype(np.float32), value_preds=rs.randn(T + 1, N, A, 1).astype(np.float32),
                masks=(rs.rand(T + 1, N, A, 1) > 0.2).astype(np.float32),
                bad_masks=(rs.rand(T + 1, N, A, 1) > 0.15).astype(np.float32),
                active_masks=(rs.rand(T + 1, N, A, 1) > 0.1).astype(np.float32),
                next_value=rs.randn(N, A, 1).astype(np.float32),
                vn_state=np.array([0.3e-3, 2.5e-3, 1.2e-3], np.float32))
    for k, v in base.items():
        out["rand_" + k] = v
    for use_gae in (True, False):
        for proper in (False, True):
            for use_vn in (False, True):
                cfg = _cfg(["--use_gae", str(use_gae), "--use_proper_time_limits", str(proper), "--use_valuenorm",
                            str(use_vn)], N, T)
                cfg.num_agents = A
                buf = NormalReplayBuffer(cfg, A, Box(-1, 1, (4,)), Discrete(2), data_client=None).data
                buf.rewards[:] = base["rewards"]
                buf.value_preds[:] = base["value_preds"]
                buf.masks[:] = base["masks"]
                buf.bad_masks[:] = base["bad_masks"]
                vn = None
                if use_vn:
                    vn = ValueNorm(1)
                    vn.running_mean[:] = float(base["vn_state"][0])
                    vn.running_mean_sq[:] = float(base["vn_state"][1])
                    vn.debiasing_term.fill_(float(base["vn_state"][2]))
                buf.compute_returns(base["next_value"].copy(), vn)
                tag = "rand_g%d_p%d_v%d" % (use_gae, proper, use_vn)
                out[tag + "_returns"] = buf.returns.copy()
                out[tag + "_value_preds"] = buf.value_preds.copy()
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **out)
    print("gae.npz", len(out))


def _train_case(name, argv, obs_dim, act_space_fn, N=8, T=12, use_masks=False, seed=0, a2c=False, share=False):
    """Fill a reference buffer with a hand-driven rollout of the reference module, then run
    PPOAlgorithm.train and record everything needed to replay it."""
    from gymnasium.spaces import Box
    from openrl.algorithms.ppo import PPOAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    cfg = _cfg(argv, N, T)
    cfg.seed = seed
    act_space = act_space_fn()
    obs_space = Box(-np.inf, np.inf, (obs_dim,))
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=share, rank=0, world_size=1)
    if share:  # PolicyValueNetwork: one network, one optimizer (ppo_module.py:58-69)
        out = {"theta_m0": _flat(module.models["model"])}
    else:
        out = {"theta_p0": _flat(module.models["policy"]), "theta_c0": _flat(module.models["critic"])}
    buffer = NormalReplayBuffer(cfg, 1, obs_space, act_space, data_client=None)
    if a2c:
        from openrl.algorithms.a2c import A2CAlgorithm

        algo = A2CAlgorithm(cfg, module, agent_num=1)
        out["a2c"] = np.array(1)
    else:
        algo = PPOAlgorithm(cfg, module, agent_num=1)
    rs = np.random.RandomState(100 + seed)
    obs = rs.randn(N, 1, obs_dim).astype(np.float32)
    K = act_space.n if act_space.__class__.__name__ == "Discrete" else 0
    am0 = None
    if use_masks:
        am0 = (rs.rand(N, 1, K) > 0.3).astype(np.float32)
        am0[..., 0] = 1.0
    buffer.init_buffer(obs.copy(), action_masks=am0)
    algo.prep_rollout()
    d = buffer.data
    for step in range(T):  # onpolicy_driver.py:159-192
        with torch.no_grad():
            value, action, logp, rs_a, rs_c = module.get_actions(
                d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                d.get_batch_data("rnn_states", step), d.get_batch_data("rnn_states_critic", step),
                d.get_batch_data("masks", step), action_masks=d.get_batch_data("action_masks", step))
        split = lambda x: np.array(np.split(x.detach().cpu().numpy(), N))
        values, actions, logps = split(value), split(action), split(logp)
        obs = rs.randn(N, 1, obs_dim).astype(np.float32)
        rewards = rs.rand(N, 1, 1).astype(np.float32)
        dones = rs.rand(N, 1) < 0.15
        dones_env = np.all(dones, axis=1)
        masks = np.ones((N, 1, 1), np.float32)
        masks[dones_env] = 0.0
        active = np.ones((N, 1, 1), np.float32)
        active[dones] = 0.0
        active[dones_env] = 1.0
        if step % 5 == 3:  # exercise active_masks == 0 (only reachable for multi-agent envs in the driver)
            active[rs.randint(N)] = 0.0
        bad = np.ones((N, 1, 1), np.float32)
        amask = None
        if use_masks:
            amask = (rs.rand(N, 1, K) > 0.3).astype(np.float32)
            amask[..., 0] = 1.0
        buffer.insert(obs, split(rs_a), split(rs_c), actions, logps, values, rewards, masks, active_masks=active,
                      bad_masks=bad, action_masks=amask)
    with torch.no_grad():
        nv = module.get_values(d.get_batch_data("critic_obs", -1), np.concatenate(d.rnn_states_critic[-1]),
                               np.concatenate(d.masks[-1]))
    next_values = np.array(np.split(nv.detach().cpu().numpy(), N))
    vn = module.get_critic_value_normalizer()
    buffer.compute_returns(next_values, vn)
    for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
              "bad_masks", "active_masks"):
        out["buf_" + f] = getattr(d, f).copy()
    if d.action_masks is not None:
        out["buf_action_masks"] = d.action_masks.copy()
    if act_space.__class__.__name__ == "Tuple":  # the mixed branch: (Box(cd), Discrete(n))
        out["mixed"] = np.array([act_space[0].shape[0], act_space[1].n])
    out["next_values"] = next_values
    # the update: RNG state at entry decides the permutations
    torch.manual_seed(1234 + seed)
    algo.prep_training()
    info = algo.train(d)
    out["train_info"] = np.array([float(info.get(k, 0.0)) for k in
                                  ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                   "ratio")], np.float64)
    if share:
        out["theta_m1"] = _flat(module.models["model"])
    else:
        out["theta_p1"] = _flat(module.models["policy"])
        out["theta_c1"] = _flat(module.models["critic"])
    if vn is not None:
        out["vn_state1"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                                    np.float32)
    out["argv"] = np.array(argv, dtype=object) if False else np.array(" ".join(argv))
    out["perm_seed"] = np.array(1234 + seed)
    # deterministic action / value probe on fresh weights is covered by theta_p0 + oracle; record a probe on theta_p1
    probe = rs.randn(16, obs_dim).astype(np.float32)
    algo.prep_rollout()
    with torch.no_grad():
        pm = None
        if use_masks:
            pm = (rs.rand(16, K) > 0.3).astype(np.float32)
            pm[:, 0] = 1.0
            out["probe_masks"] = pm
        Hh = cfg.hidden_size
        v, a, lp, _, _ = module.get_actions(probe, probe, np.zeros((16, 1, Hh), np.float32),
                                            np.zeros((16, 1, Hh), np.float32), np.ones((16, 1), np.float32),
                                            action_masks=pm, deterministic=True)
    out["probe_obs"], out["probe_values"] = probe, v.numpy()
    out["probe_actions"], out["probe_logp"] = a.numpy().astype(np.float32), lp.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: float(info[k]) for k in info})


def _train_case_mat(name, argv, obs_dim, act_space_fn, N=6, A=3, T=10, seed=0):
    """Multi-agent feed-forward buffer + the reference's ``MATAlgorithm.train`` on the MLP ``PPOModule`` (what its own
    tests/test_algorithm/test_mat_algorithm.py builds): one summed loss, ``feed_forward_generator_transformer``
    (algorithms/mat.py:21-38, buffers/replay_data.py:707-804).  A > 1, so that minibatches of (step, env) pairs with all
    their agents differ from the row-wise generator's."""
    from gymnasium.spaces import Box
    from openrl.algorithms.mat import MATAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    cfg = _cfg(argv, N, T)
    cfg.seed = seed
    act_space = act_space_fn()
    obs_space = Box(-np.inf, np.inf, (obs_dim,))
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=False, rank=0, world_size=1)
    out = {"theta_p0": _flat(module.models["policy"]), "theta_c0": _flat(module.models["critic"]), "mat": np.array(1),
           "agents": np.array(A)}
    buffer = NormalReplayBuffer(cfg, A, obs_space, act_space, data_client=None)
    algo = MATAlgorithm(cfg, module, agent_num=A)
    rs = np.random.RandomState(300 + seed)
    obs = rs.randn(N, A, obs_dim).astype(np.float32)
    buffer.init_buffer(obs.copy())
    algo.prep_rollout()
    d = buffer.data
    for step in range(T):  # onpolicy_driver.py:159-192
        with torch.no_grad():
            value, action, logp, rs_a, rs_c = module.get_actions(
                d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                d.get_batch_data("rnn_states", step), d.get_batch_data("rnn_states_critic", step),
                d.get_batch_data("masks", step), action_masks=d.get_batch_data("action_masks", step))
        split = lambda x: np.array(np.split(x.detach().cpu().numpy(), N))
        values, actions, logps = split(value), split(action), split(logp)
        obs = rs.randn(N, A, obs_dim).astype(np.float32)
        rewards = rs.rand(N, A, 1).astype(np.float32)
        dones = rs.rand(N, A) < 0.2
        if step % 4 == 1:
            dones[rs.randint(N)] = True  # a whole env done: masks 0, active 1 (onpolicy_driver.py:118-140)
        dones_env = np.all(dones, axis=1)
        masks = np.ones((N, A, 1), np.float32)
        masks[dones_env] = 0.0
        active = np.ones((N, A, 1), np.float32)
        active[dones] = 0.0
        active[dones_env] = 1.0
        bad = np.ones((N, A, 1), np.float32)
        buffer.insert(obs, split(rs_a), split(rs_c), actions, logps, values, rewards, masks, active_masks=active,
                      bad_masks=bad, action_masks=None)
    with torch.no_grad():
        nv = module.get_values(d.get_batch_data("critic_obs", -1), np.concatenate(d.rnn_states_critic[-1]),
                               np.concatenate(d.masks[-1]))
    next_values = np.array(np.split(nv.detach().cpu().numpy(), N))
    vn = module.get_critic_value_normalizer()
    buffer.compute_returns(next_values, vn)
    for f in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "returns", "rewards", "masks",
              "bad_masks", "active_masks"):
        out["buf_" + f] = getattr(d, f).copy()
    if d.action_masks is not None:
        out["buf_action_masks"] = d.action_masks.copy()
    out["next_values"] = next_values
    torch.manual_seed(4321 + seed)
    algo.prep_training()
    info = algo.train(d)
    out["train_info"] = np.array([float(info.get(k, 0.0)) for k in
                                  ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                   "ratio")], np.float64)
    out["theta_p1"] = _flat(module.models["policy"])
    out["theta_c1"] = _flat(module.models["critic"])
    if vn is not None:
        out["vn_state1"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                                    np.float32)
    out["argv"] = np.array(" ".join(argv))
    out["perm_seed"] = np.array(4321 + seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: float(info[k]) for k in info})


def actlayer_multidiscrete_case():
    """ACTLayer with a MultiDiscrete space (act.py:26-34, 60-72, 136-151): per-component Categoricals over the same
    features.  The reference's ReplayData cannot hold MultiDiscrete actions (np.zeros((T, N, A, act_space.shape)),
    replay_data.py:161-168, raises TypeError), so parity for this head is pinned at the ACTLayer level: deterministic
    actions, log-probs of given actions, the (detached) entropy, and the gradient of sum(log-probs) w.r.t. the
    features and the head parameters."""
    from gymnasium.spaces import MultiDiscrete
    from openrl.modules.networks.utils.act import ACTLayer

    torch.manual_seed(21)
    nvec, H, B = [3, 2, 5], 64, 24
    layer = ACTLayer(MultiDiscrete(nvec), H, True, 0.01)
    rs = np.random.RandomState(21)
    for lin in layer.action_outs:  # non-trivial weights (the constructor's gain 0.01 makes every head almost uniform)
        lin.linear.weight.data.copy_(torch.tensor(0.4 * rs.randn(*lin.linear.weight.shape).astype(np.float32)))
        lin.linear.bias.data.copy_(torch.tensor(0.1 * rs.randn(*lin.linear.bias.shape).astype(np.float32)))
    x = torch.tensor(rs.randn(B, H).astype(np.float32), requires_grad=True)
    actions = torch.tensor(np.stack([rs.randint(0, k, B) for k in nvec], 1).astype(np.int64))
    active = torch.tensor((rs.rand(B, 1) > 0.2).astype(np.float32))
    logp, ent = layer.evaluate_actions(x, actions, None, active)
    logp.sum().backward()
    det_a, det_lp = layer(x.detach(), None, True)
    out = dict(nvec=np.array(nvec), x=x.detach().numpy(), actions=actions.numpy().astype(np.float32),
               active=active.numpy(), logp=logp.detach().numpy(), entropy=np.array(float(ent)),
               dx=x.grad.numpy(), det_actions=det_a.numpy().astype(np.float32), det_logp=det_lp.detach().numpy())
    for i, lin in enumerate(layer.action_outs):
        out["W%d" % i], out["b%d" % i] = lin.linear.weight.detach().numpy(), lin.linear.bias.detach().numpy()
        out["dW%d" % i], out["db%d" % i] = lin.linear.weight.grad.numpy(), lin.linear.bias.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "actlayer_multidiscrete.npz"), **out)
    print("actlayer_multidiscrete.npz", float(ent))


def actlayer_mixed_case():
    """ACTLayer with a Tuple(Box(2), Discrete(n)) space - the "mixed" branch (act.py:33-43 construction, 46-63 forward,
    126-147 evaluate_actions with its hard-coded ``split((2, 1))`` and 0.0025 / 0.01 entropy weights): joint log-prob of
    given actions, the combined entropy with and without active masks, deterministic actions, and the gradient of
    sum(log-prob) + entropy w.r.t. the features and the head parameters."""
    from gymnasium.spaces import Box, Discrete, Tuple
    from openrl.modules.networks.utils.act import ACTLayer

    torch.manual_seed(23)
    cd, n, H, B = 2, 5, 64, 24
    layer = ACTLayer(Tuple((Box(-1, 1, (cd,)), Discrete(n))), H, True, 0.01)
    rs = np.random.RandomState(23)
    g, c = layer.action_outs[0], layer.action_outs[1]
    g.fc_mean.weight.data.copy_(torch.tensor(0.3 * rs.randn(cd, H).astype(np.float32)))
    g.fc_mean.bias.data.copy_(torch.tensor(0.1 * rs.randn(cd).astype(np.float32)))
    g.logstd._bias.data.copy_(torch.tensor(0.2 * rs.randn(cd, 1).astype(np.float32)))
    c.linear.weight.data.copy_(torch.tensor(0.4 * rs.randn(n, H).astype(np.float32)))
    c.linear.bias.data.copy_(torch.tensor(0.1 * rs.randn(n).astype(np.float32)))
    x = torch.tensor(rs.randn(B, H).astype(np.float32), requires_grad=True)
    actions = torch.tensor(np.concatenate([rs.randn(B, cd), rs.randint(0, n, (B, 1))], 1).astype(np.float32))
    active = torch.tensor((rs.rand(B, 1) > 0.2).astype(np.float32))
    logp, ent = layer.evaluate_actions(x, actions, None, active)
    (logp.sum() + 3.0 * ent).backward()
    with torch.no_grad():
        _, ent_nomask = layer.evaluate_actions(x.detach(), actions, None, None)
        det_a, det_lp = layer(x.detach(), None, True)
    out = dict(shape=np.array([cd, n]), x=x.detach().numpy(), actions=actions.numpy(), active=active.numpy(),
               logp=logp.detach().numpy(), entropy=np.array(float(ent)), entropy_nomask=np.array(float(ent_nomask)),
               dx=x.grad.numpy(), det_actions=det_a.numpy().astype(np.float32), det_logp=det_lp.numpy(),
               Wm=g.fc_mean.weight.detach().numpy(), bm=g.fc_mean.bias.detach().numpy(),
               logstd=g.logstd._bias.detach().numpy().reshape(-1), Wc=c.linear.weight.detach().numpy(),
               bc=c.linear.bias.detach().numpy(), dWm=g.fc_mean.weight.grad.numpy(), dbm=g.fc_mean.bias.grad.numpy(),
               dlogstd=g.logstd._bias.grad.numpy().reshape(-1), dWc=c.linear.weight.grad.numpy(),
               dbc=c.linear.bias.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "actlayer_mixed.npz"), **out)
    print("actlayer_mixed.npz", float(ent), float(ent_nomask))


def state_dict_case():
    """``state_dict()`` of the reference's own PolicyNetwork / ValueNetwork / PolicyValueNetwork (the content of the
    checkpoints ``RLAgent.save`` pickles, rl_agent.py:187-213, rl_module.py:155-192) for three configurations, plus a
    deterministic probe through ``PPOModule.get_actions`` - the engine must load these dicts by key and reproduce
    the probe."""
    from gymnasium.spaces import Box, Discrete
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    out = {}
    confs = {"default": ([], 4, Discrete(2), False),
             "general": (["--hidden_size", "32", "--layer_N", "2", "--activation_id", "0",
                          "--use_feature_normalization", "true"], 6, Box(-1, 1, (3,)), False),
             "shared": (["--use_share_model", "true"], 5, Discrete(3), True)}
    rs = np.random.RandomState(77)
    for tag, (argv, D, act, share) in confs.items():
        cfg = _cfg(argv, 4, 8)
        cfg.seed = 77
        set_seed(cfg.seed)
        obs_space = Box(-np.inf, np.inf, (D,))
        module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act,
                           share_model=share, rank=0, world_size=1)
        for name, model in module.models.items():
            # move every tensor off its initial value so that a key mix-up cannot go unnoticed
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.add_(torch.tensor(np.abs(np.asarray(0.05 * rs.randn(*p_.shape), np.float32)) if p_.dim() == 0
                                         else np.asarray(0.05 * rs.randn(*p_.shape), np.float32)))
            for k, v in model.state_dict().items():
                out["%s/%s/%s" % (tag, name, k)] = v.detach().numpy().copy()
        probe = rs.randn(12, D).astype(np.float32)
        H_ = cfg.hidden_size
        with torch.no_grad():
            v, a, lp, _, _ = module.get_actions(probe, probe, np.zeros((12, 1, H_), np.float32),
                                                np.zeros((12, 1, H_), np.float32), np.ones((12, 1), np.float32),
                                                deterministic=True)
        out[tag + "/argv"] = np.array(" ".join(argv))
        out[tag + "/probe_obs"], out[tag + "/probe_values"] = probe, v.numpy()
        out[tag + "/probe_actions"], out[tag + "/probe_logp"] = a.numpy().astype(np.float32), lp.numpy()
    np.savez_compressed(os.path.join(OUT, "state_dicts.npz"), **out)
    print("state_dicts.npz", len(out))


def _train_case_full(name, argv, N, T, D, n_act, seed=0):
    """A full-size update (BASELINE.json configs[1]: 4096 envs x 128 steps, obs 4, Discrete(2), ppo_epoch 10, one
    minibatch): the buffer comes from ``oracle.fixtures.synth_update_buffer`` (regenerated from the seed by the
    test), only the reference's outputs are stored.  ``ReplayData.compute_returns`` + ``PPOAlgorithm.train``
    (algorithms/ppo.py:383-458, buffers/replay_data.py:320-423, 553-646) are the reference's own."""
    from gymnasium.spaces import Box, Discrete
    from openrl.algorithms.ppo import PPOAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    from .fixtures import synth_update_buffer

    cfg = _cfg(argv, N, T)
    cfg.seed = seed
    obs_space, act_space = Box(-np.inf, np.inf, (D,)), Discrete(n_act)
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=False, rank=0, world_size=1)
    out = {"theta_p0": _flat(module.models["policy"]), "theta_c0": _flat(module.models["critic"])}
    buffer = NormalReplayBuffer(cfg, 1, obs_space, act_space, data_client=None)
    algo = PPOAlgorithm(cfg, module, agent_num=1)
    d = buffer.data
    src = synth_update_buffer(1000 + seed, N, T, D, n_act)
    d.policy_obs[:] = src["policy_obs"]
    d.critic_obs[:] = src["policy_obs"]
    for f in ("rewards", "value_preds", "masks", "active_masks", "bad_masks", "actions", "action_log_probs",
              "action_masks"):
        getattr(d, f)[:] = src[f]
    vn = module.get_critic_value_normalizer()
    buffer.compute_returns(src["next_value"].copy(), vn)
    out["returns_probe_idx"] = np.array([[0, 0], [T // 2, N // 3], [T - 1, N - 1], [17, 5], [T - 2, 11]])
    out["returns_probe"] = np.array([d.returns[t, n, 0, 0] for t, n in out["returns_probe_idx"]], np.float32)
    out["returns_sum"] = np.array(d.returns[:-1].astype(np.float64).sum())
    torch.manual_seed(1234 + seed)
    algo.prep_training()
    info = algo.train(d)
    out["train_info"] = np.array([float(info.get(k, 0.0)) for k in
                                  ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                   "ratio")], np.float64)
    out["theta_p1"] = _flat(module.models["policy"])
    out["theta_c1"] = _flat(module.models["critic"])
    out["vn_state1"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                                np.float32)
    out["argv"] = np.array(" ".join(argv))
    out["perm_seed"] = np.array(1234 + seed)
    out["shape"] = np.array([N, T, D, n_act, 1000 + seed])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: float(info[k]) for k in info})


def _train_case_full_general(name, argv, N, T, Dp, Dc, kind, n_act, A=1, legal_masks=False, recurrent=False, seed=0):
    """Full-size updates at the other BASELINE.json shapes - configs[2] (1024 x 200, obs 17, Box(6)), configs[4]
    (4096 x 200, obs 18, Discrete(9) + random legal-move masks) and configs[3] (2048 envs x 3 agents x 25, Dict obs
    18/54, Discrete(5), GRU, chunks of 2 through ``recurrent_generator``).  Same recipe as ``_train_case_full``: the
    buffer is regenerated from a seed (``oracle.fixtures.synth_update_buffer_general``), ``compute_returns`` and
    ``PPOAlgorithm.train`` are the reference's own (algorithms/ppo.py:383-458, buffers/replay_data.py:320-423,
    553-646, 1062-1258), only outputs are stored."""
    from gymnasium.spaces import Box, Dict as DictSpace, Discrete
    from openrl.algorithms.ppo import PPOAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    from .fixtures import synth_update_buffer_general

    argv = (["--use_recurrent_policy", "true"] if recurrent else []) + argv
    cfg = _cfg(argv, N, T)
    cfg.num_agents = A
    cfg.seed = seed
    box = lambda d: Box(-np.inf, np.inf, (d,))
    obs_space = box(Dp) if Dp == Dc else DictSpace({"policy": box(Dp), "critic": box(Dc)})
    act_space = Discrete(n_act) if kind == "discrete" else Box(-1, 1, (n_act,))
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=False, rank=0, world_size=1)
    out = {"theta_p0": _flat(module.models["policy"]), "theta_c0": _flat(module.models["critic"])}
    buffer = NormalReplayBuffer(cfg, A, obs_space, act_space, data_client=None)
    algo = PPOAlgorithm(cfg, module, agent_num=A)
    d = buffer.data
    src = synth_update_buffer_general(2000 + seed, N, T, Dp, Dc, kind, n_act, A, legal_masks,
                                      cfg.hidden_size if recurrent else 0)
    d.policy_obs[:] = src["policy_obs"]  # Dict{"policy","critic"} spaces: one plain array per side (replay_data.py:70-71)
    d.critic_obs[:] = src["critic_obs"]
    fields = ["rewards", "value_preds", "masks", "active_masks", "bad_masks", "actions", "action_log_probs"]
    if kind == "discrete":
        fields.append("action_masks")
    if recurrent:
        fields += ["rnn_states", "rnn_states_critic"]
    for f in fields:
        getattr(d, f)[:] = src[f]
    vn = module.get_critic_value_normalizer()
    buffer.compute_returns(src["next_value"].copy(), vn)
    out["returns_probe_idx"] = np.array([[0, 0, 0], [T // 2, N // 3, A - 1], [T - 1, N - 1, 0], [17, 5, A // 2],
                                         [T - 2, 11, 0]])
    out["returns_probe"] = np.array([d.returns[t, n, a, 0] for t, n, a in out["returns_probe_idx"]], np.float32)
    out["returns_sum"] = np.array(d.returns[:-1].astype(np.float64).sum())
    torch.manual_seed(1234 + seed)
    algo.prep_training()
    info = algo.train(d)
    out["train_info"] = np.array([float(info.get(k, 0.0)) for k in
                                  ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                   "ratio")], np.float64)
    out["theta_p1"] = _flat(module.models["policy"])
    out["theta_c1"] = _flat(module.models["critic"])
    out["vn_state1"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                                np.float32)
    out["argv"] = np.array(" ".join(argv))
    out["perm_seed"] = np.array(1234 + seed)
    out["shape"] = np.array([N, T, Dp, Dc, n_act, A, 2000 + seed, int(legal_masks), int(recurrent)])
    out["kind"] = np.array(kind)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: float(info[k]) for k in info})


def _train_case_recurrent(name, argv, Dp, Dc, act_space_fn, N=6, A=2, T=7, seed=0, naive=False, share=False):
    """use_recurrent_policy: hand-driven rollout (rnn states zeroed on env-done, onpolicy_driver.py:91-108),
    then PPOAlgorithm.train with recurrent_generator (ppo.py:363-372, replay_data.py:1062-1258)."""
    from gymnasium.spaces import Box, Dict as DictSpace
    from openrl.algorithms.ppo import PPOAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    # naive: use_naive_recurrent_policy -> the same RNNLayer towers, trained on whole-trajectory sequences through
    # naive_recurrent_generator (ppo.py:365-372, replay_data.py:806-960)
    rnn_flag = ["--use_naive_recurrent_policy", "true"] if naive else ["--use_recurrent_policy", "true"]
    cfg = _cfg(rnn_flag + argv, N, T)
    cfg.num_agents = A
    cfg.seed = seed
    act_space = act_space_fn()
    if Dp == Dc:
        obs_space = Box(-np.inf, np.inf, (Dp,))
    else:
        obs_space = DictSpace({"policy": Box(-np.inf, np.inf, (Dp,)), "critic": Box(-np.inf, np.inf, (Dc,))})
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=share, rank=0, world_size=1)
    if share:  # PolicyValueNetwork with its own RNNLayer (policy_value_network.py:85-91): one network, one optimizer
        out = {"theta_m0": _flat(module.models["model"])}
    else:
        out = {"theta_p0": _flat(module.models["policy"]), "theta_c0": _flat(module.models["critic"])}
    buffer = NormalReplayBuffer(cfg, A, obs_space, act_space, data_client=None)
    algo = PPOAlgorithm(cfg, module, agent_num=A)
    rs = np.random.RandomState(200 + seed)

    def draw_obs():
        if Dp == Dc:
            return rs.randn(N, A, Dp).astype(np.float32)
        return {"policy": rs.randn(N, A, Dp).astype(np.float32), "critic": rs.randn(N, A, Dc).astype(np.float32)}

    buffer.init_buffer(draw_obs())
    algo.prep_rollout()
    d = buffer.data
    H = cfg.hidden_size
    for step in range(T):
        with torch.no_grad():
            value, action, logp, rs_a, rs_c = module.get_actions(
                d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                d.get_batch_data("rnn_states", step), d.get_batch_data("rnn_states_critic", step),
                d.get_batch_data("masks", step), action_masks=d.get_batch_data("action_masks", step))
        split = lambda x: np.array(np.split(x.detach().cpu().numpy(), N))
        values, actions, logps, rnn_a, rnn_c = split(value), split(action), split(logp), split(rs_a), split(rs_c)
        rewards = rs.rand(N, A, 1).astype(np.float32)
        dones = rs.rand(N, A) < 0.2
        dones[rs.randint(N)] = True
        dones_env = np.all(dones, axis=1)
        rnn_a[dones_env] = 0.0
        rnn_c[dones_env] = 0.0
        masks = np.ones((N, A, 1), np.float32)
        masks[dones_env] = 0.0
        active = np.ones((N, A, 1), np.float32)
        active[dones] = 0.0
        active[dones_env] = 1.0
        bad = np.ones((N, A, 1), np.float32)
        buffer.insert(draw_obs(), rnn_a, rnn_c, actions, logps, values, rewards, masks, active_masks=active,
                      bad_masks=bad, action_masks=None)
    with torch.no_grad():
        nv = module.get_values(d.get_batch_data("critic_obs", -1), np.concatenate(d.rnn_states_critic[-1]),
                               np.concatenate(d.masks[-1]))
    next_values = np.array(np.split(nv.detach().cpu().numpy(), N))
    vn = module.get_critic_value_normalizer()
    buffer.compute_returns(next_values, vn)
    for f in ("policy_obs", "critic_obs"):
        v = getattr(d, f)
        out["buf_" + f] = (v["policy" if f == "policy_obs" else "critic"] if isinstance(v, dict) else v).copy()
    for f in ("actions", "action_log_probs", "value_preds", "returns", "rewards", "masks", "bad_masks",
              "active_masks", "rnn_states", "rnn_states_critic"):
        out["buf_" + f] = getattr(d, f).copy()
    if d.action_masks is not None:
        out["buf_action_masks"] = d.action_masks.copy()
    out["next_values"] = next_values
    torch.manual_seed(4321 + seed)
    algo.prep_training()
    info = algo.train(d)
    out["train_info"] = np.array([float(info[k]) for k in
                                  ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm",
                                   "ratio")], np.float64)
    if share:
        out["theta_m1"] = _flat(module.models["model"])
    else:
        out["theta_p1"] = _flat(module.models["policy"])
        out["theta_c1"] = _flat(module.models["critic"])
    if vn is not None:
        out["vn_state1"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                                    np.float32)
    out["argv"] = np.array(" ".join(rnn_flag + argv))
    out["perm_seed"] = np.array(4321 + seed)
    # deterministic probe with non-trivial states and a zero mask on theta1
    B = 12
    pp, pc = rs.randn(B, Dp).astype(np.float32), rs.randn(B, Dc).astype(np.float32)
    sw = (cfg.recurrent_N, cfg.rnn_hidden_size)  # [recurrent_N, H] (GRU) or [recurrent_N, 2 H] (LSTM: h | c)
    ha, hc = (0.5 * rs.randn(B, *sw)).astype(np.float32), (0.5 * rs.randn(B, *sw)).astype(np.float32)
    pm = np.ones((B, 1), np.float32)
    pm[::3] = 0.0
    algo.prep_rollout()
    with torch.no_grad():
        if Dp == Dc:
            pp = pc
        v, a, lp, ha1, hc1 = module.get_actions(pc, pp, ha, hc, pm, deterministic=True)
    out["probe_policy_obs"], out["probe_critic_obs"], out["probe_h"], out["probe_hc"], out["probe_masks"] = pp, pc, ha, hc, pm
    out["probe_values"], out["probe_actions"], out["probe_logp"] = v.numpy(), a.numpy().astype(np.float32), lp.numpy()
    out["probe_h1"], out["probe_hc1"] = ha1.numpy(), hc1.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: float(info[k]) for k in info})


def mpe_case():
    """Trajectories of the reference's MPE simple_spread: its own ``World`` / ``Scenario`` classes (core.py,
    scenarios/simple_spread.py - numpy only) stepped with the action decoding of multiagent_env.py:268-310 restated
    by hand (MultiAgentEnv itself needs gymnasium.core / seeding, absent here)."""
    import importlib
    import types

    ref = "/root/reference/openrl/envs"
    for name, path in (("openrl.envs", ref), ("openrl.envs.mpe", ref + "/mpe"),
                       ("openrl.envs.mpe.scenarios", ref + "/mpe/scenarios")):
        if name not in sys.modules:  # package shells: skip the __init__ files that pull gymnasium wrappers in
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    Scenario = importlib.import_module("openrl.envs.mpe.scenarios.simple_spread").Scenario
    out = {}
    rs = np.random.RandomState(11)
    E, S = 6, 30
    pos0, lm0, acts = np.zeros((E, 3, 2)), np.zeros((E, 3, 2)), rs.randint(0, 5, (E, S, 3))
    traj_pos, traj_vel = np.zeros((E, S, 3, 2)), np.zeros((E, S, 3, 2))
    traj_rew, traj_obs = np.zeros((E, S, 3)), np.zeros((E, S, 3, 18))
    for e in range(E):
        sc = Scenario()
        world = sc.make_world()
        sc.reset_world(world, np.random.default_rng(100 + e))
        if e >= 3:  # start some worlds crowded so that the contact forces are exercised
            for i, ag in enumerate(world.agents):
                ag.state.p_pos = np.array([0.05 * i, 0.12 * i]) + 0.01 * rs.randn(2)
        pos0[e] = np.stack([a.state.p_pos for a in world.agents])
        lm0[e] = np.stack([l.state.p_pos for l in world.landmarks])
        for s in range(S):
            for i, ag in enumerate(world.agents):  # _set_action, discrete_action_space branch
                onehot = np.zeros(5)
                onehot[acts[e, s, i]] = 1
                ag.action.u = np.zeros(2)
                ag.action.c = np.zeros(2)
                ag.action.u[0] += onehot[1] - onehot[2]
                ag.action.u[1] += onehot[3] - onehot[4]
                ag.action.u *= 5.0
            world.step()
            traj_pos[e, s] = np.stack([a.state.p_pos for a in world.agents])
            traj_vel[e, s] = np.stack([a.state.p_vel for a in world.agents])
            ind = [sc.reward(a, world) for a in world.agents]
            traj_rew[e, s] = np.sum(ind)  # shared_reward (multiagent_env.py:191-194)
            traj_obs[e, s] = np.stack([sc.observation(a, world) for a in world.agents])
    out.update(pos0=pos0, lm0=lm0, actions=acts, pos=traj_pos, vel=traj_vel, rewards=traj_rew, obs=traj_obs)
    np.savez_compressed(os.path.join(OUT, "mpe_spread.npz"), **out)
    print("mpe_spread.npz", traj_rew.mean())


def perm_case():
    from torch.utils.data.sampler import BatchSampler, SubsetRandomSampler

    out = {}
    for seed, M, nmb in ((0, 10, 1), (3, 96, 4), (11, 1000, 3)):
        torch.manual_seed(seed)
        mbs = M // nmb
        idx = [np.array(b) for b in BatchSampler(SubsetRandomSampler(range(M)), mbs, drop_last=True)]
        out["perm_s%d_M%d_n%d" % (seed, M, nmb)] = np.stack(idx)
    np.savez_compressed(os.path.join(OUT, "perm.npz"), **out)
    print("perm.npz")


def main():
    """``python -m oracle.gen_golden [name ...]`` - without names every fixture is regenerated."""
    ref_stubs.install()
    os.makedirs(OUT, exist_ok=True)
    from gymnasium.spaces import Box, Discrete

    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    if want("gae"):
        gae_cases()
    if want("perm"):
        perm_case()
    if want("mpe_spread"):
        mpe_case()
    if want("state_dicts"):
        state_dict_case()
    if want("actlayer_multidiscrete"):
        actlayer_multidiscrete_case()
    if want("actlayer_mixed"):
        actlayer_mixed_case()
    cases = {
        "train_discrete": lambda n: _train_case(n, ["--ppo_epoch", "3", "--num_mini_batch", "2"], 4, lambda: Discrete(2)),
        "train_discrete_masks": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--use_adv_normalize", "true", "--use_huber_loss"], 6,
            lambda: Discrete(5), use_masks=True, seed=1),
        "train_gaussian": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--lr", "7e-4", "--critic_lr", "7e-4"], 5,
            lambda: Box(-1, 1, (3,)), seed=2),
        "train_novn_proper": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--use_valuenorm", "false", "--use_proper_time_limits",
                "true", "--dual_clip_ppo", "true", "--use_clipped_value_loss", "--use_value_active_masks", "false",
                "--use_policy_active_masks"], 4, lambda: Discrete(2), seed=3),
        # use_popart without ValueNorm (the reference's tests/test_buffer/test_generator.py combination): v_out is a PopArt
        # layer whose update / normalize are never called on this path; model.parameters() gains 4 frozen entries
        "train_popart": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--use_popart", "true", "--use_valuenorm", "false"], 5,
            lambda: Discrete(3), seed=16),
        # A2CAlgorithm (algorithms/a2c.py): policy-gradient loss, one minibatch per epoch
        "train_a2c": lambda n: _train_case(n, ["--ppo_epoch", "3", "--num_mini_batch", "4"], 4, lambda: Discrete(3),
                                           seed=7, a2c=True),
        # general towers (MLPBase / MLPLayer beyond the default: mlp.py:8-46,100-180) and the shared PolicyValueNetwork
        "train_gen_h128_l2_tanh_fn": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--hidden_size", "128", "--layer_N", "2", "--activation_id",
                "0", "--use_feature_normalization", "true"], 6, lambda: Discrete(3), seed=11),
        "train_gen_elu_box": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--hidden_size", "32", "--activation_id", "3"], 5,
            lambda: Box(-1, 1, (2,)), seed=12),
        "train_gen_leaky_l3": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--layer_N", "3", "--activation_id", "2",
                "--use_adv_normalize", "true"], 4, lambda: Discrete(2), use_masks=True, seed=13),
        # A2CAlgorithm on a general tower (the loss variant through the general path's loss kernels)
        "train_gen_a2c": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "3", "--hidden_size", "48", "--layer_N", "2", "--activation_id", "0"],
            5, lambda: Discrete(4), use_masks=True, seed=17, a2c=True),
        # the mixed ACTLayer branch through a whole PPOAlgorithm.train: Tuple(Box(2), Discrete(4)) actions, one joint
        # log-prob broadcast over the 3 stored columns (-> 3 summed surrogate columns), 0.0025 / 0.01 entropy weights
        "train_gen_mixed": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2"], 5,
            lambda: __import__("gymnasium").spaces.Tuple((Box(-1, 1, (2,)), Discrete(4))), seed=18),
        # four 128-wide layers (the widest instances of the cross-layer fused update kernels, csrc/orl_gen_tower.h): the
        # shared network at hidden 128 (base + common) and layer_N 3 with ELU + feature norm + Box actions
        "train_share_h128": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--use_share_model", "true", "--hidden_size", "128"], 6,
            lambda: Discrete(3), seed=41, share=True),
        "train_gen_h128_l3_elu_fn": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--hidden_size", "128", "--layer_N", "3", "--activation_id", "3",
                "--use_feature_normalization", "true"], 9, lambda: Box(-1, 1, (2,)), seed=42),
        "train_share": lambda n: _train_case(
            n, ["--ppo_epoch", "3", "--num_mini_batch", "2", "--use_share_model", "true"], 5, lambda: Discrete(4),
            seed=14, share=True),
        "train_share_box_fn": lambda n: _train_case(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--use_share_model", "true",
                "--use_feature_normalization", "true", "--hidden_size", "48", "--layer_N", "2"], 7,
            lambda: Box(-1, 1, (3,)), seed=15, share=True),
        # MATAlgorithm on the MLP PPOModule (algorithms/mat.py): summed loss, (step, env)-pair minibatches, 3 agents
        "train_mat": lambda n: _train_case_mat(
            n, ["--ppo_epoch", "3", "--num_mini_batch", "2", "--lr", "7e-4", "--critic_lr", "7e-4"], 6, lambda: Discrete(5),
            N=6, A=3, T=10, seed=21),
        "train_mat_box": lambda n: _train_case_mat(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--use_huber_loss"], 5, lambda: Box(-1, 1, (2,)), N=5, A=2,
            T=9, seed=22),
        # recurrent (GRU) branch: T=7 is odd, so chunks of 2 straddle lanes like cfg4's T=25
        "train_recurrent": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--lr", "7e-4", "--critic_lr", "7e-4"], 18, 54,
            lambda: Discrete(5), N=6, A=3, T=7, seed=5),
        # BASELINE.json configs[1] at full size (what bench.py runs): only outputs are stored, inputs come from a seed
        "train_cfg2_full": lambda n: _train_case_full(n, ["--ppo_epoch", "10", "--num_mini_batch", "1"], 4096, 128, 4, 2),
        # the other BASELINE.json shapes at full size: the launch branches of the tower pair that only large batches take
        # (uneven CU split, back-to-back order), the ragged multi-chunk-per-wave rounds of the recurrent update
        "train_cfg3_full": lambda n: _train_case_full_general(
            n, ["--ppo_epoch", "10", "--num_mini_batch", "1"], 1024, 200, 17, 17, "box", 6, seed=3),
        "train_cfg5_full": lambda n: _train_case_full_general(
            n, ["--ppo_epoch", "10", "--num_mini_batch", "1"], 4096, 200, 18, 18, "discrete", 9, legal_masks=True, seed=5),
        "train_cfg4_full": lambda n: _train_case_full_general(
            n, ["--ppo_epoch", "10", "--num_mini_batch", "1", "--data_chunk_length", "2", "--lr", "7e-4", "--critic_lr",
                "7e-4", "--use_adv_normalize", "true"], 2048, 25, 18, 54, "discrete", 5, A=3, recurrent=True, seed=4),
        # JRPO: use_joint_action_loss -> recurrent_generator_v3 (agent axis kept) + joint ratio, critic on agent 0 only
        # (algorithms/ppo.py:254-300, buffers/replay_data.py:425-551)
        "train_recurrent_jrpo": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--use_joint_action_loss", "true", "--lr", "7e-4",
                "--critic_lr", "7e-4"], 18, 54, lambda: Discrete(5), N=6, A=3, T=8, seed=8),
        # use_naive_recurrent_policy: whole trajectories per lane (naive_recurrent_generator)
        "train_naive_recurrent": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--lr", "7e-4", "--critic_lr", "7e-4"], 7, 7,
            lambda: Discrete(4), N=5, A=2, T=6, seed=9, naive=True),
        "train_recurrent_chunk5": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--data_chunk_length", "5"], 6, 6,
            lambda: Box(-1, 1, (2,)), N=5, A=1, T=10, seed=6),
        # recurrent GENERAL towers (any hidden_size / layer_N / activation + the GRU of rnn.py): hidden 128 at the MPE
        # shape; a tanh tower with layer_N 2, feature norm, Box actions and chunks of 4 that straddle lanes (T = 10)
        "train_recurrent_gen_h128": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--hidden_size", "128", "--lr", "7e-4", "--critic_lr", "7e-4"],
            18, 54, lambda: Discrete(5), N=6, A=3, T=7, seed=31),
        "train_recurrent_gen_jrpo": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--use_joint_action_loss", "true", "--hidden_size", "96",
                "--layer_N", "2", "--lr", "7e-4", "--critic_lr", "7e-4"], 18, 54, lambda: Discrete(5), N=6, A=3, T=8, seed=33),
        # a stack of two GRU layers (recurrent_N 2) on the DEFAULT trunk: states [.., 2, H]
        "train_recurrent_gen_n2": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--recurrent_N", "2", "--data_chunk_length", "3"], 7, 7,
            lambda: Discrete(4), N=5, A=2, T=9, seed=34),
        # LSTM cells (rnn_type lstm): states [h | c] of width 2 H; one layer on a non-default trunk, two layers on the default
        "train_recurrent_gen_lstm": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--rnn_type", "lstm", "--hidden_size", "48", "--data_chunk_length",
                "3"], 7, 7, lambda: Discrete(4), N=5, A=2, T=9, seed=35),
        "train_recurrent_gen_lstm_n2": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--rnn_type", "lstm", "--recurrent_N", "2"], 6, 6,
            lambda: Box(-1, 1, (2,)), N=4, A=1, T=8, seed=36),
        # the shared PolicyValueNetwork with its RNNLayer: both passes (actor / critic states) go through one GRU
        "train_share_recurrent": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "2", "--use_share_model", "true", "--hidden_size", "40",
                "--data_chunk_length", "3"], 6, 6, lambda: Discrete(4), N=5, A=2, T=9, seed=37, share=True),
        "train_recurrent_gen_l2_tanh_fn": lambda n: _train_case_recurrent(
            n, ["--ppo_epoch", "2", "--num_mini_batch", "1", "--hidden_size", "32", "--layer_N", "2", "--activation_id",
                "0", "--use_feature_normalization", "true", "--data_chunk_length", "4"], 6, 6,
            lambda: Box(-1, 1, (2,)), N=5, A=1, T=10, seed=32),
    }
    for name, fn in cases.items():
        if want(name):
            fn(name)


if __name__ == "__main__":
    sys.exit(main())

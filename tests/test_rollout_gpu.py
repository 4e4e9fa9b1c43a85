"""Rollout-side parity on a MI355X: device envs vs their oracle restatements, fused vs stepwise rollout,
teacher-forced check of the fused rollout against the oracle towers, and end-to-end training."""
import numpy as np
import pytest
import torch

from oracle import philox as px
from oracle import ppo_oracle as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(argv=()):
    from openrl_amd.configs.config import default_cfg

    return default_cfg(list(argv))


def test_synthetic_env_matches_oracle_stream():
    from openrl_amd.envs.common import make

    N, D, L = 37, 6, 5
    env = make("SyntheticFixedStep-v0", env_num=N, obs_dim=D, episode_limit=L, seed=123, device=DEV)
    orc = po.SynthEnvOracle(N, D, 123, L)
    obs, _ = env.reset(seed=123)
    np.testing.assert_allclose(obs, orc.reset(), rtol=1e-5, atol=2e-6)
    for _ in range(12):
        o, r, d, _ = env.step(None)
        oo, rr, dd, _ = orc.step()
        np.testing.assert_allclose(o, oo, rtol=1e-5, atol=2e-6)
        assert np.array_equal(r, rr)          # rewards: integer->float conversion only, bit-exact
        assert np.array_equal(d, dd)          # done schedule: integer arithmetic, bit-exact


def test_cartpole_step_matches_gymnasium_dynamics_restatement():
    from openrl_amd.envs.common import make

    N = 512
    env = make("CartPole-v1", env_num=N, seed=7, device=DEV)
    obs0, _ = env.reset(seed=7)
    want0 = po.cartpole_reset_state(7, np.arange(N), np.zeros(N))
    np.testing.assert_allclose(obs0[:, 0], want0, rtol=0, atol=1e-7)
    rs = np.random.RandomState(0)
    state = obs0[:, 0].copy()
    for t in range(30):
        a = rs.randint(0, 2, N)
        o, r, d, _ = env.step(a.reshape(N, 1, 1).astype(np.float32))
        nxt, term = po.cartpole_step_f32(state, a)
        live = ~d[:, 0]
        np.testing.assert_allclose(o[live, 0], nxt[live], rtol=2e-5, atol=2e-6)
        assert np.array_equal(d[:, 0], term)   # no truncation within 30 steps
        assert np.all(r == 1.0)
        # auto-reset: a finished env restarts from its next keyed reset state
        state = np.where(d, o[:, 0], nxt)


def _build(env_id, N, T, seed=3, **kw):
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    cfg = _cfg(["--seed", str(seed), "--episode_length", str(T)])
    env = make(env_id, env_num=N, device=DEV, seed=seed, **kw)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)

    class _Agent:
        num_time_steps = 0

    cfg.num_env_steps = N * T
    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
    return cfg, env, net, trainer, buf, _Agent()


@pytest.mark.parametrize("env_id,kw,N,T", [
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=7), 50, 23),
    ("CartPole-v1", {}, 50, 23),
    # the headline size (bench.py: BASELINE.json configs[1]): 256 workgroups x 128 steps of the fused kernel
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=200), 4096, 128),
    # wide Gaussian head (cfg3's HalfCheetah shape): the fused kernel samples per lane from the MFMA fragment
    ("SyntheticFixedStep-v0", dict(obs_dim=17, episode_limit=9, action_space="box6"), 70, 21),
    # narrow Gaussian head (scalar head path)
    ("SyntheticFixedStep-v0", dict(obs_dim=5, episode_limit=9, action_space="box3"), 40, 11),
])
def test_fused_rollout_equals_stepwise_rollout(env_id, kw, N, T):
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver

    box = None
    if isinstance(kw.get("action_space"), str):  # "box<k>" -> Box(-1, 1, (k,))
        from openrl_amd import spaces

        box = int(kw["action_space"][3:])
        kw = dict(kw, action_space=spaces.Box(-1.0, 1.0, (box,)))
    bufs = []
    for mode in ("fused", "stepwise"):
        cfg, env, net, trainer, buf, agent = _build(env_id, N, T, **kw)
        cfg.amd_rollout_mode = mode
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf,
                             agent)
        assert drv.fused == (mode == "fused")
        drv.reset_and_buffer_init()
        drv.actor_rollout()
        drv.compute_returns()
        assert agent.num_time_steps == N * T
        bufs.append(buf.data)
    a, b = bufs
    # Same Philox streams, same towers.  The fused kernel splits a tile's GEMMs over 4 waves (identical k order per
    # output element) but hipcc contracts the LayerNorm / head arithmetic differently in the two kernels, so float
    # fields agree to a few ulp, not bitwise; a sampled action can flip only when its uniform sits on a CDF edge.
    act_a, act_b = a.actions.cpu().numpy(), b.actions.cpu().numpy()
    if box is not None:  # continuous actions: mean + std * eps with the same Philox normals, to fp32 round-off
        np.testing.assert_allclose(act_a, act_b, rtol=1e-5, atol=1e-6)
        same = np.ones_like(act_a, dtype=bool)
    else:
        same = act_a == act_b
    assert same.mean() >= 0.999, same.mean()
    if env_id.startswith("Synthetic"):  # observations do not depend on actions: every step is comparable
        for f in ("policy_obs", "rewards", "masks", "active_masks", "bad_masks"):
            assert np.array_equal(getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy()), f
        np.testing.assert_allclose(a.value_preds.cpu().numpy(), b.value_preds.cpu().numpy(), rtol=1e-5, atol=1e-6)
        lp_a, lp_b = a.action_log_probs.cpu().numpy(), b.action_log_probs.cpu().numpy()
        np.testing.assert_allclose(lp_a[same], lp_b[same], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a.returns.cpu().numpy(), b.returns.cpu().numpy(), rtol=1e-4, atol=1e-5)
    else:  # CartPole trajectories are chaotic after a flipped action: compare the first steps of every env
        for f in ("policy_obs", "value_preds", "action_log_probs", "rewards", "masks"):
            x, y = getattr(a, f).cpu().numpy()[:3], getattr(b, f).cpu().numpy()[:3]
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6, err_msg=f)


@pytest.mark.parametrize("env_id,kw,N,T", [
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=7), 50, 23),
    ("CartPole-v1", {}, 200, 40),
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=200), 4096, 128),
    ("SyntheticFixedStep-v0", dict(obs_dim=17, episode_limit=9, action_space="box6"), 1024, 50),
    ("SyntheticFixedStep-v0", dict(obs_dim=18, episode_limit=9, action_space="disc9"), 130, 20),
    ("SyntheticFixedStep-v0", dict(obs_dim=5, episode_limit=9, action_space="box2"), 40, 11),
    # degenerate sizes: fewer envs than a tile, fewer steps than the rings are deep / than there are critic waves
    ("SyntheticFixedStep-v0", dict(obs_dim=4, episode_limit=3), 3, 1),
    ("CartPole-v1", {}, 17, 2),
    ("SyntheticFixedStep-v0", dict(obs_dim=8, episode_limit=4), 33, 5),
])
def test_chain_rollout_kernel_equals_the_lockstep_kernel(env_id, kw, N, T):
    """Round 6: the policy-only chain kernel (csrc/orl_rollout2.h: head from per-wave partials of LayerNorm 2, services on
    their own waves, values from one batched critic sweep) against the round-5 lock-step kernel on the same seeds: env
    streams bit-exact, actions identical except where a uniform sits on a CDF edge, float fields to fp32 round-off.
    (HIP against HIP: the reference-side evidence of both is test_fused_rollout_equals_stepwise_rollout +
    test_fused_rollout_teacher_forced_vs_oracle_towers.)"""
    from openrl_amd import spaces
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver

    kind = kw.get("action_space")
    if isinstance(kind, str):
        k = int(kind[4:]) if kind.startswith("disc") else int(kind[3:])
        kw = dict(kw, action_space=spaces.Discrete(k) if kind.startswith("disc") else spaces.Box(-1.0, 1.0, (k,)))
    bufs, stats = [], []
    for kernel in ("chain", "lockstep"):
        cfg, env, net, trainer, buf, agent = _build(env_id, N, T, **kw)
        cfg.amd_rollout_kernel = kernel
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
        assert drv.fused
        drv.reset_and_buffer_init()
        for _ in range(2):  # the second rollout starts from the first one's env state / step counters
            drv.actor_rollout()
            drv.compute_returns()
            bufs.append({f: getattr(buf.data, f).cpu().numpy().copy() for f in
                         ("actions", "policy_obs", "rewards", "masks", "active_masks", "bad_masks", "value_preds",
                          "action_log_probs", "returns")})
            buf.data.after_update()
        stats.append((env.env_state.cpu().numpy().copy(), env.ep_stats.cpu().numpy().copy()))
    for a, b in ((bufs[0], bufs[2]), (bufs[1], bufs[3])):
        if isinstance(kind, str) and kind.startswith("box"):
            np.testing.assert_allclose(a["actions"], b["actions"], rtol=1e-5, atol=2e-6)
            same = np.ones_like(a["actions"], dtype=bool)
        else:
            same = a["actions"] == b["actions"]
        assert same.mean() >= 0.999, same.mean()
        if env_id.startswith("Synthetic"):
            for f in ("policy_obs", "rewards", "masks", "active_masks", "bad_masks"):
                assert np.array_equal(a[f], b[f]), f
            # (the chain kernel's values come from the bf16-split sweep - fp32 accuracy, another rounding pattern)
            np.testing.assert_allclose(a["value_preds"], b["value_preds"], rtol=2e-5, atol=3e-6)
            np.testing.assert_allclose(a["action_log_probs"][same], b["action_log_probs"][same], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(a["returns"], b["returns"], rtol=1e-4, atol=1e-5)
        else:  # CartPole: an env whose action flipped once is on another trajectory from there on, and the physics amplifies
            #    the few-ulp differences of the two kernels' step arithmetic: the first steps tightly, the rest loosely
            ok_env = same.all(axis=(0, 2, 3))
            assert ok_env.mean() >= 0.97, ok_env.mean()
            for f in ("policy_obs", "rewards", "masks", "value_preds", "action_log_probs"):
                np.testing.assert_allclose(a[f][:4], b[f][:4], rtol=2e-5, atol=2e-6, err_msg=f)
                np.testing.assert_allclose(a[f][:, ok_env], b[f][:, ok_env], rtol=2e-3, atol=2e-4, err_msg=f)
    if env_id.startswith("Synthetic"):
        assert np.array_equal(stats[0][0], stats[1][0]) and np.array_equal(stats[0][1], stats[1][1])


def test_fused_rollout_teacher_forced_vs_oracle_towers():
    N, T = 64, 16
    cfg, env, net, trainer, buf, agent = _build("SyntheticFixedStep-v0", N, T, obs_dim=5, episode_limit=6)
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver

    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
    drv.reset_and_buffer_init()
    drv.actor_rollout()
    d = buf.data
    mod = net.module
    pspec, cspec = po.TowerSpec(5, 2, po.HEAD_CATEGORICAL), po.TowerSpec(5, 1, po.HEAD_VALUE)
    tp, tc = mod.models["policy"].theta.cpu(), mod.models["critic"].theta.cpu()
    obs = d.policy_obs.cpu().numpy()
    agree = []
    for t in range(T):
        x, _, _, _ = px.philox4x32_10(mod.act_seed, np.arange(N, dtype=np.uint32), 0, t, 0)
        u = px.u01(x).reshape(N, 1)
        v, a, lp = po.get_actions(pspec, tp, cspec, tc, obs[t, :, 0], obs[t, :, 0], None, False, u)
        np.testing.assert_allclose(d.value_preds[t, :, 0].cpu().numpy(), v, rtol=1e-4, atol=1e-5)
        same = d.actions[t, :, 0, 0].cpu().numpy() == a[:, 0]
        agree.append(same.mean())
        np.testing.assert_allclose(d.action_log_probs[t, :, 0].cpu().numpy()[same], lp[same], rtol=1e-4, atol=1e-5)
    assert np.mean(agree) >= 0.995
    # env side: obs slots follow the oracle env stream, masks follow the done schedule
    orc = po.SynthEnvOracle(N, 5, env.seed, 6)
    np.testing.assert_allclose(obs[0], orc.reset(), rtol=1e-5, atol=2e-6)
    for t in range(T):
        oo, rr, dd, _ = orc.step()
        np.testing.assert_allclose(obs[t + 1], oo, rtol=1e-5, atol=2e-6)
        assert np.array_equal(d.rewards[t].cpu().numpy(), rr)
        assert np.array_equal(d.masks[t + 1].cpu().numpy()[:, :, 0], np.where(dd, 0.0, 1.0).astype(np.float32))


def test_agent_train_api_and_callbacks():
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent
    from openrl_amd.utils.callbacks import BaseCallback

    class Count(BaseCallback):
        def _on_step(self):
            assert "obs" in self.locals
            return True

    cfg = _cfg(["--episode_length", "20", "--ppo_epoch", "2"])
    env = make("CartPole-v1", env_num=9, device=DEV)
    net = Net(env, cfg=cfg, device=DEV)
    agent = Agent(net)
    cb = Count()
    agent.train(total_time_steps=9 * 20 * 3, callback=cb)
    assert cb.n_calls * 9 == agent.num_time_steps == 9 * 20 * 3
    obs, info = env.reset()
    action, _ = agent.act(obs, deterministic=True)
    assert action.shape == (9, 1, 1)
    o, r, d, i = env.step(action)
    assert o.shape == (9, 1, 4) and r.shape == (9, 1, 1) and d.shape == (9, 1)


def test_callback_factory_checkpoint_and_stop_on_max_episodes(tmp_path):
    """cfg.callbacks-style specs (examples/cartpole/callbacks.yaml) through CallbackFactory: training stops after
    max_episodes per env, checkpoints land where asked and load back."""
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent
    from openrl_amd.utils.callbacks import CallbackFactory

    cfg = _cfg(["--episode_length", "16", "--ppo_epoch", "1"])
    env = make("CartPole-v1", env_num=8, device=DEV)
    agent = Agent(Net(env, cfg=cfg, device=DEV))
    cbs = CallbackFactory.get_callbacks([
        {"id": "ProgressBarCallback"},
        {"id": "StopTrainingOnMaxEpisodes", "args": {"max_episodes": 3, "verbose": 1}},
        {"id": "CheckpointCallback", "args": {"save_freq": 20, "save_path": str(tmp_path / "ck"), "name_prefix": "ppo"}}])
    agent.train(total_time_steps=8 * 16 * 400, callback=cbs)
    stop, ck = cbs.callbacks[1], cbs.callbacks[2]
    assert 8 * 3 <= stop.n_episodes < 8 * 3 + 8 and agent.num_time_steps < 8 * 16 * 400  # stopped early
    assert len(ck.saved) == agent.num_time_steps // 8 // 20 and len(ck.saved) >= 1
    theta = agent.net.module.models["policy"].theta.clone()
    agent.net.module.models["policy"].theta.zero_()
    agent.load(ck.saved[-1])
    assert agent.net.module.models["policy"].theta.abs().sum() > 0
    with pytest.raises(ValueError):
        CallbackFactory.get_callback({"id": "SelfplayAPI"})  # the self-play HTTP service callbacks are not built
    del theta


def test_eval_callback_with_reward_threshold_stops_training(tmp_path):
    """examples/cartpole/callbacks.yaml's EvalCallback + StopTrainingOnRewardThreshold on a device-resident eval env."""
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent
    from openrl_amd.utils.callbacks import CallbackFactory

    cfg = _cfg(["--episode_length", "32", "--ppo_epoch", "4", "--lr", "1e-3", "--critic_lr", "1e-3"])
    env = make("CartPole-v1", env_num=64, device=DEV)
    agent = Agent(Net(env, cfg=cfg, device=DEV))
    cb = CallbackFactory.get_callback({"id": "EvalCallback", "args": {
        "eval_env": {"id": "CartPole-v1", "env_num": 16, "device": DEV}, "n_eval_episodes": 2, "eval_freq": 64,
        "best_model_save_path": str(tmp_path / "best"), "verbose": 0,
        "callbacks_on_new_best": [{"id": "StopTrainingOnRewardThreshold", "args": {"reward_threshold": 150.0}}]}})
    agent.train(total_time_steps=64 * 32 * 400, callback=cb)
    assert len(cb.evaluations) >= 1 and cb.best_mean_reward >= 150.0, cb.evaluations
    assert agent.num_time_steps < 64 * 32 * 400, "training did not stop at the threshold"
    assert all(e[2] >= 32 for e in cb.evaluations) and (tmp_path / "best").exists()
    assert cb.evaluations[-1][1] == cb.best_mean_reward  # the evaluation that crossed the threshold ended the run


def test_cartpole_learns_like_the_reference_recipe(tmp_path):
    """The reference's behavioural test (tests/test_examples/test_train_cartpole.py:39-54): default cfg, 9 envs,
    20 000 steps, then a deterministic rollout.  Its own >= 450 bar is on the FIRST termination among the 9
    envs and needs real gymnasium; the oracle port of the reference's maths (pinned on golden vectors) reaches
    first-termination lengths 292 / 500 / 381 / 343 / 294 for seeds 0-4 on this env restatement, this engine
    245 / 500 / 351 / 388 / 444 - same distribution.  Bar here: mean deterministic episode
    return over the 9 envs >= 350 of 500 (a random policy scores ~22)."""
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

    cfg = _cfg(["--seed", "1"])
    env = make("CartPole-v1", env_num=9, device=DEV)
    agent = Agent(Net(env, cfg=cfg, device=DEV))
    agent.train(total_time_steps=20000)
    agent.save(tmp_path / "ckpt")
    agent.load(tmp_path / "ckpt")
    env2 = make("CartPole-v1", env_num=9, device=DEV, seed=99)
    agent.set_env(env2)
    obs, _ = env2.reset(seed=99)
    total = np.zeros(9)
    alive = np.ones(9, bool)
    for _ in range(500):
        action, _ = agent.act(obs, deterministic=True)
        obs, r, done, _ = env2.step(action)
        total += r[:, 0, 0] * alive
        alive &= ~done[:, 0]
        if not alive.any():
            break
    assert total.mean() >= 350, total


@pytest.mark.parametrize("obs_dim,act", [(64, ("discrete", 16)), (33, ("box", 16)), (1, ("discrete", 2))])
def test_extreme_tower_shapes_train_end_to_end(obs_dim, act):
    """The widest (obs 64, 16 outputs) and narrowest towers the kernels admit through make / PPONet / PPOAgent.train on
    the synthetic env: fused rollout, GAE, update - finite parameters, permutation and buffers consistent."""
    from openrl_amd import spaces
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

    space = spaces.Discrete(act[1]) if act[0] == "discrete" else spaces.Box(-1.0, 1.0, (act[1],))
    cfg = _cfg(["--episode_length", "12", "--ppo_epoch", "2", "--amd_perm_mode", "device"])
    env = make("SyntheticFixedStep-v0", env_num=40, obs_dim=obs_dim, action_space=space, episode_limit=5, device=DEV)
    agent = Agent(Net(env, cfg=cfg, device=DEV))
    agent.train(total_time_steps=40 * 12 * 3)
    assert agent.driver.fused and agent.num_time_steps == 40 * 12 * 3
    d = agent.driver.buffer.data
    assert torch.isfinite(d.returns).all() and torch.isfinite(d.action_log_probs).all()
    if act[0] == "discrete":
        assert d.actions.min() >= 0 and d.actions.max() <= act[1] - 1 and len(torch.unique(d.actions)) > 1
    for m in agent.net.module.models.values():
        assert torch.isfinite(m.theta).all() and torch.isfinite(m.grad).all()


@pytest.mark.parametrize("mode", ["stepwise", "auto"])
@pytest.mark.parametrize("obs_dim,n_act", [(64, 16), (18, 5), (1, 2)])
def test_extreme_recurrent_tower_shapes_train_end_to_end(obs_dim, n_act, mode):
    """Same for the recurrent (GRU) towers: stepwise rollout (hipGraph from the second iteration) or, since round 3, the
    fused recurrent rollout on the single-agent device envs (amd_rollout_mode auto) + chunked BPTT update."""
    from openrl_amd import spaces
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

    cfg = _cfg(["--episode_length", "12", "--ppo_epoch", "2", "--amd_perm_mode", "device", "--use_recurrent_policy", "true",
                "--data_chunk_length", "4", "--amd_rollout_mode", mode])
    env = make("SyntheticFixedStep-v0", env_num=40, obs_dim=obs_dim, action_space=spaces.Discrete(n_act), episode_limit=5,
               device=DEV)
    agent = Agent(Net(env, cfg=cfg, device=DEV))
    agent.train(total_time_steps=40 * 12 * 3)
    assert agent.driver.fused == (mode == "auto") and agent.num_time_steps == 40 * 12 * 3
    d = agent.driver.buffer.data
    assert torch.isfinite(d.returns).all() and d.rnn_states.abs().max() > 0
    assert d.actions.min() >= 0 and d.actions.max() <= n_act - 1
    for m in agent.net.module.models.values():
        assert torch.isfinite(m.theta).all() and torch.isfinite(m.grad).all()


@pytest.mark.parametrize("env_id,argv", [("CartPole-v1", []), ("SyntheticFixedStep-v0", ["--hidden_size", "128", "--layer_N", "2"]),
                                         ("CartPole-v1", ["--hidden_size", "32", "--use_share_model", "true"])])
def test_graph_captured_rollout_on_the_base_device_envs_equals_eager(env_id, argv):
    """The synthetic / CartPole env's step counter has a device part (orl_env_step_dev), so a stepwise rollout of these
    envs is captured into a hipGraph too (fixed towers in stepwise mode, general towers always): buffers, weights and
    the env's own counter are bit-identical to the eager stepwise run."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    def run(use_graph):
        N, T, iters = 80, 20, 4
        cfg = default_cfg(["--seed", "5", "--episode_length", str(T), "--ppo_epoch", "2", "--amd_perm_mode", "device",
                           "--amd_use_graph", str(use_graph), "--amd_rollout_mode", "stepwise", "--log_interval", "1000000"] + argv)
        kw = dict(obs_dim=6, episode_limit=7) if env_id.startswith("Synthetic") else {}
        env = make(env_id, env_num=N, device=DEV, **kw)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        cfg.num_env_steps = N * T * iters

        class _Agent:
            num_time_steps = 0

        agent = _Agent()
        trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
        buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
        drv.reset_and_buffer_init()
        for i in range(iters):
            drv.episode = i
            drv._inner_loop()
        assert (drv._graph is not None) == use_graph and agent.num_time_steps == N * T * iters
        assert env.global_step == T * iters
        d = buf.data
        keep = {k: getattr(d, k).clone() for k in ("policy_obs", "actions", "action_log_probs", "value_preds", "rewards",
                                                   "masks", "returns")}
        theta = next(iter(net.module.models.values())).theta.clone()
        return keep, theta, net.module.rng_step, env.env_state.clone()

    a, tha, ra, ea = run(True)
    b, thb, rb, eb = run(False)
    assert ra == rb and torch.equal(ea, eb)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(tha, thb)

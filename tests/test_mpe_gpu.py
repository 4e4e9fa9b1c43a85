"""Device-resident MPE simple_spread (csrc/orl_mpe.hip) vs the reference trajectories / the oracle restatement, and
BASELINE config 4 end to end: make("simple_spread") + PPONet + PPOAgent.train with use_recurrent_policy."""
import numpy as np
import pytest
import torch

from oracle import mpe_oracle as mo
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state(pos, vel, lm, step=0.0):
    E = pos.shape[0]
    st = np.zeros((E, 24), np.float32)
    st[:, 0:6], st[:, 6:12], st[:, 12:18], st[:, 18] = pos.reshape(E, 6), vel.reshape(E, 6), lm.reshape(E, 6), step
    return torch.tensor(st, device=DEV)


def test_mpe_step_matches_reference_trajectories_teacher_forced():
    """One device step from every golden state (the reference integrates in float64; fp32 tolerance stated here)."""
    from openrl_amd import ops_rnn

    g = H.load_golden("mpe_spread")
    E, S = g["actions"].shape[:2]
    assert ops_rnn.mpe_state_width() == 24
    obs_p, obs_c = torch.zeros(E, 3, 18, device=DEV), torch.zeros(E, 3, 54, device=DEV)
    rew, done = torch.zeros(E, 3, device=DEV), torch.zeros(E, 3, dtype=torch.uint8, device=DEV)
    for s in range(S):
        pos = g["pos0"] if s == 0 else g["pos"][:, s - 1]
        vel = np.zeros_like(pos) if s == 0 else g["vel"][:, s - 1]
        st = _state(pos, vel, g["lm0"], step=float(s % 20))
        ops_rnn.mpe_step(st, None, torch.tensor(g["actions"][:, s], dtype=torch.float32, device=DEV), obs_p, obs_c, rew,
                         done, E, 0, 25)
        stn = st.cpu().numpy()
        np.testing.assert_allclose(stn[:, 0:6].reshape(E, 3, 2), g["pos"][:, s], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(stn[:, 6:12].reshape(E, 3, 2), g["vel"][:, s], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(obs_p.cpu().numpy(), g["obs"][:, s], rtol=1e-4, atol=2e-5)
        # rewards are discontinuous at contact (|d - 0.3| < 1e-6 could flip a collision count): none in the fixture
        np.testing.assert_allclose(rew.cpu().numpy(), g["rewards"][:, s], rtol=1e-5, atol=1e-5)
        oc = obs_c.cpu().numpy()
        assert np.array_equal(oc[:, 0], obs_p.cpu().numpy().reshape(E, 54)) and np.array_equal(oc[:, 1], oc[:, 0])
        assert not done.any() and np.all(stn[:, 18] == float(s % 20) + 1)


def test_mpe_reset_done_and_autoreset():
    from openrl_amd.envs.common import make

    N = 70
    env = make("simple_spread", env_num=N, seed=4)
    assert env.agent_num == 3 and env.parallel_env_num == N and env.action_space.n == 5
    obs = env.reset_device(seed=4)
    for n in (0, 13, 69):
        ag, lm = mo.device_reset_positions(4, n, 0)
        p, c = mo.observations(ag.astype(np.float64), np.zeros((3, 2)), lm.astype(np.float64))
        np.testing.assert_allclose(obs["policy"][n].cpu().numpy(), p, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(obs["critic"][n].cpu().numpy(), c, rtol=1e-6, atol=1e-6)
    a = torch.zeros(N, 3, 1, device=DEV)
    total = torch.zeros(N, device=DEV)
    for t in range(25):
        obs, rew, done = env.step_device(a)
        total += rew[:, 0, 0]
        assert rew.shape == (N, 3, 1) and done.shape == (N, 3)
        assert bool(done.all()) == (t == 24)
    # auto-reset: the observation returned with done is the first one of episode 1 (sync_venv.py:217-222)
    ag, lm = mo.device_reset_positions(4, 13, 1)
    p, _ = mo.observations(ag.astype(np.float64), np.zeros((3, 2)), lm.astype(np.float64))
    np.testing.assert_allclose(obs["policy"][13].cpu().numpy(), p, rtol=1e-6, atol=1e-6)
    s = env.episode_statistics()
    assert s["episodes_finished"] == N
    np.testing.assert_allclose(s["episode_return_mean"], total.mean().item(), rtol=1e-5)
    # no-op actions from rest: nothing moves, so the reward is constant over the episode
    o2, r2, d2, _ = env.step(np.zeros((N, 3, 1)))
    assert o2["policy"].shape == (N, 3, 18) and r2.shape == (N, 3, 1) and d2.dtype == bool


def test_config4_end_to_end_recurrent_mappo_on_device_mpe():
    """examples/mpe/mpe_ppo.yaml (use_recurrent_policy, episode_length 25, lr 7e-4) on the device MPE env."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent

    N, T = 128, 25
    cfg = default_cfg(["--seed", "0", "--lr", "7e-4", "--critic_lr", "7e-4", "--episode_length", str(T),
                       "--use_recurrent_policy", "true", "--use_valuenorm", "true", "--use_adv_normalize", "true",
                       "--ppo_epoch", "5", "--log_interval", "1"])
    env = make("simple_spread", env_num=N)
    net = PPONet(env, cfg=cfg, device=DEV)
    agent = PPOAgent(net)
    agent.train(total_time_steps=3 * N * T)
    drv = agent.driver
    assert drv.fused and drv.fused_rnn and agent.num_time_steps == 3 * N * T  # orl_rnn_rollout_fused
    d = drv.buffer.data
    assert d.policy_obs.shape == (T + 1, N, 3, 18) and d.critic_obs.shape == (T + 1, N, 3, 54)
    # episodes are exactly one rollout long: slot 0 (= previous slot T) starts an episode: mask 0, states 0
    assert torch.all(d.masks[0] == 0) and torch.all(d.masks[1:T] == 1)
    assert torch.all(d.rnn_states[0] == 0) and d.rnn_states[1:T].abs().max() > 0
    assert torch.isfinite(d.rewards).all() and d.rewards.max() < 0
    for m in net.module.models.values():
        assert torch.isfinite(m.theta).all()
    st = env.episode_statistics()
    assert st["episodes_finished"] == 3 * N and -400 < st["episode_return_mean"] < -50


def test_recurrent_mappo_learns_on_device_mpe():
    """Learning sanity for config 4: the shared episode reward of simple_spread improves markedly within 150
    iterations of recurrent MAPPO (tools/mpe_learning_curve.py: -218 -> -123 after 600 iterations of 1024 envs)."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T, iters = 512, 25, 150
    cfg = default_cfg(["--seed", "0", "--lr", "7e-4", "--critic_lr", "7e-4", "--episode_length", str(T),
                       "--use_recurrent_policy", "true", "--use_valuenorm", "true", "--use_adv_normalize", "true",
                       "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = make("simple_spread", env_num=N, device=DEV)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
    cfg.num_env_steps = N * T * iters

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=3, device=DEV)
    buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=DEV)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    curve = []
    for i in range(iters):
        drv.episode = i
        drv._inner_loop()
        curve.append(float(buf.data.rewards[:, :, 0, 0].sum(0).mean()))
    first, last = np.mean(curve[:3]), np.mean(curve[-10:])
    assert last > first + 40.0, (first, last)


@pytest.mark.parametrize("recurrent", ["true", "false"])
def test_graph_captured_rollout_equals_eager_rollout(recurrent):
    """amd_use_graph: the stepwise rollout is captured once into a hipGraph and replayed, with the Philox step counter
    on the device (orl_act_rng_offset).  Same kernels, same counters -> the buffers are bit-identical to eager."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    def run(use_graph):
        N, T, iters = 96, 25, 4
        cfg = default_cfg(["--seed", "3", "--episode_length", str(T), "--use_recurrent_policy", recurrent, "--ppo_epoch", "2",
                           "--amd_perm_mode", "device", "--amd_use_graph", str(use_graph), "--amd_rollout_mode", "stepwise",
                           "--log_interval", "1000000"])
        env = make("simple_spread", env_num=N, device=DEV, seed=3)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        cfg.num_env_steps = N * T * iters

        class _Agent:
            num_time_steps = 0

        agent = _Agent()
        trainer = PPOAlgorithm(cfg, net.module, agent_num=3, device=DEV)
        buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
        drv.reset_and_buffer_init()
        for i in range(iters):
            drv.episode = i
            drv._inner_loop()
        assert (drv._graph is not None) == use_graph and agent.num_time_steps == N * T * iters
        d = buf.data
        return {k: getattr(d, k).clone() for k in ("policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds",
                                                     "rewards", "masks", "active_masks", "rnn_states", "returns")}, \
            net.module.models["policy"].theta.clone(), net.module.rng_step

    a, tha, ra = run(True)
    b, thb, rb = run(False)
    assert ra == rb
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(tha, thb)


def test_chase_error_word_is_sticky_across_rollouts():
    """a critic-wait timeout recorded by rollout k must survive rollout k + 1's launch (which used to
    memset the error word together with the step counters), so that a host that polls late - DeviceErrorWatch reuses one
    pinned word - still sees it.  The timeout is simulated by setting the device word between two rollouts."""
    from openrl_amd import _native as nat
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T = 48, 6
    cfg = default_cfg(["--seed", "3", "--episode_length", str(T), "--use_recurrent_policy", "true", "--ppo_epoch", "1",
                       "--amd_perm_mode", "device", "--amd_use_graph", "false", "--amd_rollout_mode", "fused",
                       "--amd_rnn_rollout_chase", "true", "--log_interval", "1000000"])
    env = make("simple_spread", env_num=N, device=DEV, seed=3)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
    cfg.num_env_steps = N * T * 3

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=3, device=DEV)
    buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=DEV)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    drv.actor_rollout()
    torch.cuda.synchronize()
    assert drv._chase_flags is not None and int(drv._chase_flags[-1]) == 0
    drv._chase_flags[-1] = 1            # "a critic workgroup timed out in this rollout"
    drv.learner_update()
    buf.data.after_update()
    with pytest.raises(nat.NativeError, match="timed out"):
        drv.actor_rollout()             # re-launch: counters cleared, error word kept; post() copies it, poll() raises
        torch.cuda.synchronize()
        assert int(drv._chase_flags[-1]) == 1 and int(drv._chase_flags[:-1].min()) == T
        drv._chase_watch.poll(wait=True)


@pytest.mark.parametrize("N,chase", [(96, "true"), (50, "true"), (96, "false"), (50, "false"), (2048, "true")])
def test_fused_recurrent_rollout_equals_stepwise_rollout(N, chase):
    """orl_rnn_rollout_fused (policy + MPE world in one launch, critic sweep in a second) against the stepwise
    T x {orl_rnn_act_step, orl_mpe_step, orl_buffer_insert_rnn} rollout: same per-tile arithmetic and Philox counters,
    so actions / rewards / masks agree exactly and the float fields to fp32 round-off; 4 iterations with updates in
    between, so slot 0 hand-over (after_update), auto-reset and the bootstrap value are covered.  N = 50 leaves a
    ragged last tile of 2 worlds."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    def run(mode):
        T, iters = 25, 4
        cfg = default_cfg(["--seed", "3", "--episode_length", str(T), "--use_recurrent_policy", "true", "--ppo_epoch", "2",
                           "--amd_perm_mode", "device", "--amd_use_graph", "false", "--amd_rollout_mode", mode,
                           "--amd_rnn_rollout_chase", chase, "--log_interval", "1000000"])
        env = make("simple_spread", env_num=N, device=DEV, seed=3)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        cfg.num_env_steps = N * T * iters

        class _Agent:
            num_time_steps = 0

        agent = _Agent()
        trainer = PPOAlgorithm(cfg, net.module, agent_num=3, device=DEV)
        buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, agent)
        assert drv.fused == (mode == "fused")
        drv.reset_and_buffer_init()
        snaps = []
        for i in range(iters):
            drv.episode = i
            drv.actor_rollout()
            drv.learner_update()  # compute_returns (consumes the fused launch's bootstrap value) + train
            d = buf.data
            snaps.append({k: getattr(d, k).clone() for k in (
                "policy_obs", "critic_obs", "actions", "action_log_probs", "value_preds", "rewards", "masks",
                "active_masks", "bad_masks", "rnn_states", "rnn_states_critic", "returns")})
            snaps[-1]["env_obs_p"], snaps[-1]["env_obs_c"] = env.obs["policy"].clone(), env.obs["critic"].clone()
            snaps[-1]["env_state"], snaps[-1]["ep_stats"] = env.env_state.clone(), env.ep_stats.clone()
            buf.after_update()
        assert agent.num_time_steps == N * T * iters
        if drv._chase_flags is not None:  # the last word is set when a critic's bounded wait timed out
            assert int(drv._chase_flags[-1]) == 0 and int(drv._chase_flags[:-1].min()) == T
        return snaps, net.module.models["policy"].theta.clone(), net.module.rng_step, env.global_step

    a, tha, ra, ga = run("fused")
    b, thb, rb, gb = run("stepwise")
    assert ra == rb and ga == gb
    exact = ("actions", "rewards", "masks", "active_masks", "bad_masks", "policy_obs", "critic_obs", "env_obs_p", "env_obs_c",
             "env_state", "ep_stats")
    # iteration 0 starts from identical weights; later iterations inherit fp32 round-off through the updates, where a
    # sampled action may flip - compare them on the first iteration exactly and on all iterations statistically
    for k in exact:
        assert torch.equal(a[0][k], b[0][k]), k
    for k in ("action_log_probs", "value_preds", "rnn_states", "rnn_states_critic", "returns"):
        torch.testing.assert_close(a[0][k], b[0][k], rtol=2e-5, atol=2e-6, msg=k)
    for i in range(1, len(a)):
        same = (a[i]["actions"] == b[i]["actions"]).float().mean().item()
        assert same > 0.99, (i, same)
        assert torch.equal(a[i]["masks"], b[i]["masks"])
    torch.testing.assert_close(tha, thb, rtol=1e-3, atol=1e-4)

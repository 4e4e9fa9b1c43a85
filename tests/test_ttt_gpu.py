"""Device-resident tic-tac-toe (csrc/orl_ttt.hip) vs the oracle restatement, and BASELINE config 5's shape end to end:
make("tictactoe_v3") + PPONet + PPOAgent with legal-move masks that never leave the device."""
import numpy as np
import pytest
import torch

from oracle import ttt_oracle as to

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ttt_step_matches_oracle_bit_for_bit():
    from openrl_amd.envs.common import make

    N, S = 300, 40
    env = make("tictactoe_v3", env_num=N, seed=11, device=DEV)
    assert env.agent_num == 1 and env.action_space.n == 9 and env.observation_space.shape == (18,)
    obs = env.reset_device(seed=11)
    games = [to.Game(11, n) for n in range(N)]
    rs = np.random.RandomState(0)
    n_done = 0
    for s in range(S):
        o, m = obs.cpu().numpy().reshape(N, 18), env.action_mask_device.cpu().numpy().reshape(N, 9)
        for n in (0, 7, N - 1) if s else range(N):
            assert np.array_equal(o[n], games[n].obs()) and np.array_equal(m[n], games[n].mask())
        acts = np.array([rs.choice(np.flatnonzero(m[n])) for n in range(N)])
        illegal = rs.rand(N) < 0.03  # now and then an occupied cell (if there is one): the mover loses
        for n in np.flatnonzero(illegal):
            occ = np.flatnonzero(m[n] == 0)
            if occ.size:
                acts[n] = occ[0]
        obs, rew, done = env.step_device(torch.tensor(acts, dtype=torch.float32, device=DEV).view(N, 1, 1))
        r, d = rew.cpu().numpy().reshape(N), done.cpu().numpy().reshape(N)
        for n in range(N):
            wr, wd = games[n].step(int(acts[n]))
            assert (r[n], bool(d[n])) == (wr, wd), (s, n)
        n_done += int(d.sum())
    o, m = obs.cpu().numpy().reshape(N, 18), env.action_mask_device.cpu().numpy().reshape(N, 9)
    for n in range(N):
        assert np.array_equal(o[n], games[n].obs()) and np.array_equal(m[n], games[n].mask())
    st = env.episode_statistics()
    assert st["episodes_finished"] == n_done and n_done > 2 * N


@pytest.mark.parametrize("mode", ["fused", "graph", "eager"])
def test_config5_end_to_end_masks_stay_on_device(mode):
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T, iters = 256, 8, 3
    cfg = default_cfg(["--seed", "2", "--episode_length", str(T), "--ppo_epoch", "2", "--amd_perm_mode", "device",
                       "--amd_use_graph", str(mode == "graph"), "--amd_rollout_mode",
                       "fused" if mode == "fused" else "stepwise", "--log_interval", "1000000"])
    env = make("tictactoe_v3", env_num=N, device=DEV, seed=2)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
    cfg.num_env_steps = N * T * iters

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    for i in range(iters):
        drv.episode = i
        drv._inner_loop()
        d = buf.data
        acts = d.actions[:, :, 0, 0].long()                       # [T, N]
        legal = d.action_masks[:T, :, 0, :].gather(-1, acts.unsqueeze(-1)).squeeze(-1)
        # slot 0 already holds the NEXT rollout's first mask (after_update copied slot T there): check slots 1..T-1
        assert torch.all(legal[1:] == 1), "a masked (occupied) cell was sampled"
        # masks are consistent with the stored observations: empty <=> neither plane set
        occ = d.policy_obs[:, :, 0, 0::2] + d.policy_obs[:, :, 0, 1::2]
        assert torch.equal(1.0 - occ, d.action_masks[:, :, 0, :])
        assert torch.all(d.rewards.abs() <= 1) and torch.isfinite(d.returns).all()
    assert (drv._graph is not None) == (mode == "graph") and drv.fused == (mode == "fused")
    for m in net.module.models.values():
        assert torch.isfinite(m.theta).all()


def _ttt_rollouts(mode, kernel, N, T, iters=2):
    """`iters` rollouts of tic-tac-toe vs the random opponent through the driver; returns the buffers after each, the episode
    statistics and the env state"""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    cfg = default_cfg(["--seed", "5", "--episode_length", str(T), "--ppo_epoch", "1", "--amd_perm_mode", "device",
                       "--amd_use_graph", "false", "--amd_rollout_mode", mode, "--log_interval", "1000000",
                       "--amd_rollout_kernel", kernel])
    env = make("tictactoe_v3", env_num=N, device=DEV, seed=5)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
    cfg.num_env_steps = N * T * iters

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    out = []
    for i in range(iters):
        drv.episode = i
        drv.actor_rollout()
        drv.compute_returns()
        d = buf.data
        out.append({k: getattr(d, k).clone() for k in ("policy_obs", "actions", "action_log_probs", "value_preds",
                                                       "rewards", "masks", "action_masks", "returns")})
        drv.buffer.after_update()
    return out, env.ep_stats.clone(), env.env_state.clone()


def _same_rollouts(ra, rb):
    (fa, sa, ea), (fb, sb, eb) = ra, rb
    for a, b in zip(fa, fb):
        for k in ("policy_obs", "actions", "rewards", "masks", "action_masks"):
            assert torch.equal(a[k], b[k]), k
        # the kernels' float arithmetic is contracted differently by hipcc (and the chain kernel's critic runs fc2 on the
        # bf16 x 3 split): a few ulp, not bitwise
        for k in ("action_log_probs", "value_preds", "returns"):
            torch.testing.assert_close(a[k], b[k], rtol=2e-5, atol=3e-6, msg=k)
    assert torch.equal(sa, sb) and torch.equal(ea, eb)


@pytest.mark.parametrize("kernel", ["chain", "lockstep"])
def test_fused_rollout_equals_stepwise_rollout_on_tictactoe(kernel):
    """The in-kernel game of orl_rollout_fused (ORL_ENV_TTT) and the stepwise orl_act_step + orl_ttt_step +
    orl_buffer_insert path fill the buffer identically (same Philox streams for sampling and for the opponent).
    `chain` = the round-6 kernel (csrc/orl_rollout2.h: the board travels as one word per row through its rings), `lockstep` =
    round 5's.  N is not a multiple of the 16-env tile."""
    _same_rollouts(_ttt_rollouts("fused", kernel, 150, 12), _ttt_rollouts("stepwise", kernel, 150, 12))


@pytest.mark.parametrize("N,T", [(3, 1), (17, 2), (33, 5), (1000, 200)])
def test_chain_rollout_equals_lockstep_rollout_on_tictactoe(N, T):
    """The two fused kernels against each other at degenerate sizes (fewer rows than a tile, one step: the rings never wrap)
    and at configuration 5's episode length (every ring wraps 25 - 50 times, thousands of auto-resets)."""
    _same_rollouts(_ttt_rollouts("fused", "chain", N, T), _ttt_rollouts("fused", "lockstep", N, T))


@pytest.mark.parametrize("sampling", ["static", "per_rollout", "per_reset"])
def test_fused_selfplay_rollout_equals_stepwise(sampling):
    """ORL_ENV_TTT_POOL inside orl_rollout_fused (both players' policies in-kernel) fills the buffer like the stepwise
    orl_act_step + orl_ttt_agent_move + orl_act_step_grouped + orl_ttt_opponent_move + orl_buffer_insert path."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    def run(mode):
        N, T = 150, 12
        cfg = default_cfg(["--seed", "5", "--episode_length", str(T), "--ppo_epoch", "1", "--amd_perm_mode", "device",
                           "--amd_use_graph", "false", "--amd_rollout_mode", mode, "--log_interval", "1000000"])
        env = make("tictactoe_v3", env_num=N, device=DEV, seed=5, opponent="pool", pool_size=3,
                   opponent_sampling=sampling)
        torch.manual_seed(4)
        env.opp_thetas.copy_(0.5 * torch.randn_like(env.opp_thetas))  # three different non-trivial opponents
        env.pushes = 3  # all three slots count as filled for the per-rollout draw
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        cfg.num_env_steps = N * T * 2

        class _Agent:
            num_time_steps = 0

        trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
        buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
        assert drv.fused == (mode == "fused")
        drv.reset_and_buffer_init()
        out = []
        for i in range(2):
            drv.episode = i
            drv.actor_rollout()
            drv.compute_returns()
            d = buf.data
            out.append({k: getattr(d, k).clone() for k in ("policy_obs", "actions", "action_log_probs", "value_preds",
                                                           "rewards", "masks", "action_masks", "returns")})
            drv.buffer.after_update()
        out[-1]["opp_index"] = env.opp_index.clone()  # per_reset: the slots the in-kernel draws ended on
        out[-1]["draws"] = torch.tensor(env._draws)
        return out, env.ep_stats.clone(), env.env_state[:, :11].clone()

    (fa, sa, ea), (fb, sb, eb) = run("fused"), run("stepwise")
    assert torch.equal(fa[-1]["opp_index"], fb[-1]["opp_index"]) and torch.equal(fa[-1]["draws"], fb[-1]["draws"])
    if sampling == "per_reset":
        assert len(torch.unique(fa[-1]["opp_index"])) == 3  # every slot is in play within a tile
    for a, b in zip(fa, fb):
        for k in ("policy_obs", "actions", "rewards", "masks", "action_masks"):
            assert torch.equal(a[k], b[k]), k
        for k in ("action_log_probs", "value_preds", "returns"):
            torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-6, msg=k)
    assert torch.equal(sa, sb) and torch.equal(ea, eb)
    assert (fa[0]["rewards"] != 0).sum() > 50  # games were decided both ways


def test_ppo_beats_the_random_opponent():
    """Learning sanity for config 5's env: the mean game result against the uniformly random opponent (0 for a random
    agent by symmetry) rises clearly within 120 iterations of masked PPO."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T, iters = 1024, 10, 120
    cfg = default_cfg(["--seed", "0", "--lr", "1e-3", "--critic_lr", "1e-3", "--episode_length", str(T),
                       "--ppo_epoch", "5", "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = make("tictactoe_v3", env_num=N, device=DEV)
    net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
    cfg.num_env_steps = N * T * iters

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    curve = []
    for i in range(iters):
        drv.episode = i
        drv._inner_loop()
        d = buf.data
        ends = (d.masks[1:] == 0).sum().clamp(min=1)
        curve.append(float(d.rewards.sum() / ends))  # mean result of the games finished in this rollout
    first, last = np.mean(curve[:3]), np.mean(curve[-10:])
    assert abs(first) < 0.25 and last > first + 0.4, (first, last)


def test_two_phase_step_matches_oracle_with_scripted_opponent():
    """orl_ttt_agent_move + orl_ttt_opponent_move (the self-play step) against the oracle, the opponent's replies drawn
    by the test from the opponent-side masks the first kernel wrote."""
    from openrl_amd import ops_rnn

    N, S, seed = 200, 30, 9
    z = lambda *s, **k: torch.zeros(*s, device=DEV, **k)
    st, eps, obs, mask = z(N, ops_rnn.ttt_state_width()), z(N, 4), z(N, 18), z(N, 9)
    oobs, omask, rew, done = z(N, 18), z(N, 9), z(N), z(N, dtype=torch.uint8)
    ops_rnn.ttt_reset(st, eps, obs, mask, N, seed)
    games = [to.Game(seed, n) for n in range(N)]
    rs = np.random.RandomState(1)
    for s in range(S):
        m = mask.cpu().numpy()
        acts = np.array([rs.choice(np.flatnonzero(m[n])) for n in range(N)], np.float32)
        ops_rnn.ttt_agent_move(st, torch.tensor(acts, device=DEV), oobs, omask, rew, done, N)
        oo, om, r1, d1 = oobs.cpu().numpy(), omask.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        opp = np.zeros(N, np.float32)
        for n in range(N):
            wr, wd, pending = games[n].agent_move(int(acts[n]))
            wo, wm = games[n].opponent_view()
            assert (r1[n], bool(d1[n])) == (wr, wd) and np.array_equal(oo[n], wo) and np.array_equal(om[n], wm), (s, n)
            opp[n] = rs.choice(np.flatnonzero(wm))
        ops_rnn.ttt_opponent_move(st, eps, torch.tensor(opp, device=DEV), obs, mask, rew, done, N, seed)
        r2, d2, o2, m2 = rew.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy(), mask.cpu().numpy()
        for n in range(N):
            wr, wd = games[n].opponent_move(int(opp[n]))
            assert (r2[n], bool(d2[n])) == (wr, wd), (s, n)
            assert np.array_equal(o2[n], games[n].obs()) and np.array_equal(m2[n], games[n].mask())
    assert float(eps[:, 3].sum()) > N  # games were finished and counted


def test_selfplay_pool_env_and_callback_end_to_end():
    """make("tictactoe_v3", opponent="pool"): opponents are frozen snapshots of the learner, refreshed by
    SelfPlayCallback through PPOAgent.train; afterwards the learner beats the uniformly random opponent."""
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent
    from openrl_amd.utils.callbacks import SelfPlayCallback

    N, T = 1024, 10
    cfg = default_cfg(["--seed", "0", "--lr", "1e-3", "--critic_lr", "1e-3", "--episode_length", str(T), "--ppo_epoch", "5",
                       "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = make("tictactoe_v3", env_num=N, device=DEV, opponent_wrappers=["RecordWinner", "OpponentPoolWrapper"],
               pool_size=3, opponent_sampling="per_rollout")
    # before any snapshot: all-zero parameters = uniform over the legal moves (log-prob = -log(#empty cells))
    env.reset_device(seed=0)
    env.step_device(torch.zeros(N, 1, 1, device=DEV))
    lp = env._opp_lp.view(-1).cpu().numpy()
    n_legal = env._opp_mask.sum(-1).cpu().numpy()
    np.testing.assert_allclose(lp, -np.log(n_legal), rtol=1e-5, atol=1e-6)
    net = PPONet(env, cfg=cfg, device=DEV)
    agent = PPOAgent(net)
    agent.train(total_time_steps=N * T * 80, callback=SelfPlayCallback(push_every=10))
    assert env.pushes == 7 and agent.driver.fused
    assert env.opp_thetas.abs().sum(dim=1).min() > 0  # every slot holds a real snapshot by now
    for m in net.module.models.values():
        assert torch.isfinite(m.theta).all()
    # evaluation against the uniformly random opponent
    ev = make("tictactoe_v3", env_num=2048, device=DEV, seed=123)
    obs = ev.reset_device(seed=123)
    for _ in range(40):
        a, _ = net.module.act(obs.view(2048, 18), None, None, action_masks=ev.action_mask_device.view(2048, 9), deterministic=True)
        obs, _, _ = ev.step_device(a.view(2048, 1, 1))
    st = ev.episode_statistics()
    assert st["episodes_finished"] > 2048 * 5 and st["episode_return_mean"] > 0.5, st


def test_act_step_pool_equals_one_launch_per_policy():
    """orl_act_step_pool (every row names its pool slot; a tile runs once per distinct slot it holds) against K plain
    orl_act_step launches over the whole batch with the rows picked afterwards: same Philox counters, same towers."""
    from openrl_amd import ops

    B, K, D, NA = 200, 4, 18, 9
    pnet = ops.net_desc(D, NA, ops.HEAD_CATEGORICAL)
    g = torch.Generator(device=DEV).manual_seed(3)
    thetas = 0.3 * torch.randn(K, ops.param_count(pnet), device=DEV, generator=g)
    obs = torch.randn(B, D, device=DEV, generator=g)
    masks = (torch.rand(B, NA, device=DEV, generator=g) > 0.3).float()
    masks[:, 0] = 1.0
    idx = torch.randint(0, K, (B,), device=DEV, generator=g).int()
    idx[:16] = 2           # a tile with a single slot
    idx[16:32] = torch.arange(16, device=DEV).int() % K  # a tile with every slot
    a, lp = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
    ops.act_step_pool(pnet, thetas, idx, obs, masks, B, False, 99, 7, 5, a, lp)
    want_a, want_lp = torch.empty_like(a), torch.empty_like(lp)
    for k in range(K):
        ak, lk = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV)
        ops.act_step(pnet, thetas[k], None, None, obs, None, masks, B, False, 99, 7, 5, None, None, ak, lk)
        sel = idx == k
        want_a[sel], want_lp[sel] = ak[sel], lk[sel]
    assert torch.equal(a, want_a)
    assert torch.equal(lp, want_lp)


@pytest.mark.parametrize("strategy", ["RandomOpponent", "LastOpponent"])
def test_per_reset_opponent_sampling(strategy):
    """opponent_sampling="per_reset" (opponent_pool_wrapper.py:37-66): an env draws a new pool slot exactly when its
    game ends - uniform over the FILLED slots (RandomOpponent) or the newest slot (LastOpponent) - and keeps it
    through the game (the stepwise step; test_fused_selfplay_rollout_equals_stepwise[per_reset] holds the fused kernel's
    in-kernel draws to the same stream).  Pools of more than 4 snapshots do not fit the fused kernel's LDS."""
    from openrl_amd.envs.common import make

    N = 2048
    env = make("tictactoe_v3", env_num=N, device=DEV, seed=9, opponent="pool", pool_size=4,
               opponent_sampling="per_reset", opponent_strategy=strategy)
    assert env.supports_fused_rollout
    assert not make("tictactoe_v3", env_num=N, device=DEV, seed=9, opponent="pool", pool_size=6,
                    opponent_sampling="per_reset").supports_fused_rollout
    g = torch.Generator(device=DEV).manual_seed(1)
    for k in range(3):  # three snapshots pushed: slots 0..2 filled, 3 empty
        env.push_opponent(0.3 * torch.randn(env.opp_thetas.shape[1], device=DEV, generator=g))
    env.reset_device(seed=9)
    idx0 = env.opp_index.clone()
    assert int(idx0.max()) <= 2
    changed_total, seen = 0, []
    for t in range(12):
        prev = env.opp_index.clone()
        am = env.action_mask_device.view(N, 9)
        a = torch.multinomial(am + 1e-6, 1, generator=g).float().view(N, 1, 1)  # a random legal move
        _, _, done = env.step_device(a)
        now = env.opp_index
        assert torch.equal(now[done.view(-1) == 0], prev[done.view(-1) == 0]), "an open game keeps its opponent"
        seen.append(now[done.view(-1) != 0].clone())
        changed_total += int((done.view(-1) != 0).sum())
    drawn = torch.cat(seen)
    assert changed_total > N and int(drawn.max()) <= 2
    if strategy == "LastOpponent":
        assert bool((drawn == 2).all())  # the newest snapshot sits in slot 2
    else:
        freq = torch.bincount(drawn, minlength=3).float() / drawn.numel()
        assert float((freq - 1 / 3).abs().max()) < 0.03, freq

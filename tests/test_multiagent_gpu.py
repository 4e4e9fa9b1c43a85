"""Multi-agent (MAPPO-style, feed-forward) and Dict{"policy","critic"} observations through the stepwise driver
with a HOST numpy env (the duck-typed VecEnv contract of examples/isaac/isaac2openrl.py:28-88), at the MPE
simple_spread dimensions of BASELINE config 4 (3 agents, policy obs 18, critic obs 54, Discrete(5))."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class ToyMultiAgentEnv:
    """N envs x A agents, random observations, per-agent dones, one bad transition per 50 steps."""

    def __init__(self, n, a=3, dp=18, dc=54, n_act=5, seed=0):
        from openrl_amd import spaces

        self.n, self.a, self.dp, self.dc = n, a, dp, dc
        self.rs = np.random.RandomState(seed)
        self.observation_space = spaces.Dict({"policy": spaces.Box(-np.inf, np.inf, (dp,)),
                                              "critic": spaces.Box(-np.inf, np.inf, (dc,))})
        self.action_space = spaces.Discrete(n_act)
        self.t = 0
        self.seen_actions = []

    parallel_env_num = property(lambda s: s.n)
    agent_num = property(lambda s: s.a)
    env_name = "toy_spread"
    use_monitor = False

    def _obs(self):
        return {"policy": self.rs.randn(self.n, self.a, self.dp).astype(np.float32),
                "critic": self.rs.randn(self.n, self.a, self.dc).astype(np.float32)}

    def reset(self, seed=None, options=None):
        return self._obs(), [{} for _ in range(self.n)]

    def step(self, actions, extra_data=None):
        assert actions.shape == (self.n, self.a, 1)
        self.seen_actions.append(actions.copy())
        self.t += 1
        dones = self.rs.rand(self.n, self.a) < 0.1
        dones[0] = True  # env 0: all agents done every step -> masks 0, active 1
        infos = [{"bad_transition": [self.t % 50 == 0 and i == 1] * self.a} for i in range(self.n)]
        return self._obs(), self.rs.rand(self.n, self.a, 1).astype(np.float32), dones, infos

    def batch_rewards(self, buffer):
        return {}

    def close(self):
        pass


def test_mappo_feedforward_stepwise_host_env_and_update_vs_oracle():
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.common import PPONet
    from openrl_amd.runners.common import PPOAgent

    N, A, T = 6, 3, 10
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "1", "--seed", "4", "--use_proper_time_limits", "true"])
    env = ToyMultiAgentEnv(N, A)
    net = PPONet(env, cfg=cfg, device=DEV)
    agent = PPOAgent(net)
    theta0 = {k: m.theta.cpu().clone() for k, m in net.module.models.items()}
    agent.train(total_time_steps=N * T)  # one rollout (stepwise: host env) + one update
    assert agent.num_time_steps == N * T and len(env.seen_actions) == T
    d = agent.driver.buffer.data
    assert d.policy_obs.shape == (T + 1, N, A, 18) and d.critic_obs.shape == (T + 1, N, A, 54)
    assert d.critic_obs is not d.policy_obs
    masks, active, bad = d.masks.cpu().numpy(), d.active_masks.cpu().numpy(), d.bad_masks.cpu().numpy()
    # after_update rolled slot T into slot 0; slots 1..T hold this rollout
    assert np.all(masks[1:, 0] == 0) and np.all(active[1:, 0] == 1)       # whole env done
    # only some agents of an env done -> masks stay 1, that agent's active mask drops to 0
    # (onpolicy_driver.py:99-133)
    assert np.all(masks[1:, 1:] == 1) and active[1:, 1:].min() == 0.0 and set(np.unique(active)) <= {0.0, 1.0}
    assert np.all(bad == 1.0)  # t never reaches 50 here
    # the engine's GAE on the buffer it filled itself vs the oracle (slots 1..T are untouched by after_update)
    g = lambda name: getattr(d, name).cpu().numpy()
    ret, _ = po.compute_returns(g("rewards"), g("value_preds"), g("masks"), g("bad_masks"), g("value_preds")[-1],
                                cfg.gamma, cfg.gae_lambda, True, True, po.ValueNormOracle())
    np.testing.assert_allclose(g("returns")[:-1], ret[:-1], rtol=1e-5, atol=1e-5)
    assert d.records.shape == (T * N * A, 84)  # 18 + 54 + act 1 + logp 1 + 4 scalars + 5 action masks, padded to 4
    for k, m in net.module.models.items():
        th = m.theta.cpu()
        assert torch.isfinite(th).all() and (th - theta0[k]).abs().max() > 0
    action, _ = agent.act({"policy": np.zeros((N, A, 18), np.float32), "critic": np.zeros((N, A, 54), np.float32)})
    assert action.shape == (N, A, 1)


def test_dict_obs_update_gradients_vs_oracle():
    """Different policy / critic observation widths (18 / 54): single full-batch update vs oracle autograd."""
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.ppo_module import PPOModule

    N, A, T, Dp, Dc, K = 10, 3, 8, 18, 54, 5
    rs = np.random.RandomState(2)
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "1"])
    cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = N, A, cfg.hidden_size
    obs_space = spaces.Dict({"policy": spaces.Box(-np.inf, np.inf, (Dp,)), "critic": spaces.Box(-np.inf, np.inf, (Dc,))})
    act_space = spaces.Discrete(K)
    torch.manual_seed(1)
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=DEV, rank=0, world_size=1)
    buf = ReplayData(cfg, A, obs_space, act_space, device=DEV)
    host = dict(policy_obs=rs.randn(T + 1, N, A, Dp).astype(np.float32),
                critic_obs=rs.randn(T + 1, N, A, Dc).astype(np.float32),
                rewards=rs.rand(T, N, A, 1).astype(np.float32),
                value_preds=(0.3 * rs.randn(T + 1, N, A, 1)).astype(np.float32),
                masks=(rs.rand(T + 1, N, A, 1) > 0.05).astype(np.float32),
                active_masks=(rs.rand(T + 1, N, A, 1) > 0.1).astype(np.float32),
                actions=rs.randint(0, K, (T, N, A, 1)).astype(np.float32),
                action_log_probs=(np.log(1.0 / K) + 0.05 * rs.randn(T, N, A, 1)).astype(np.float32))
    for k, v in host.items():
        getattr(buf, k).copy_(torch.tensor(v))
    buf.compute_returns(torch.tensor(0.3 * rs.randn(N, A, 1).astype(np.float32)), module.get_critic_value_normalizer())
    host["returns"], host["value_preds"] = buf.returns.cpu().numpy(), buf.value_preds.cpu().numpy()
    algo = PPOAlgorithm(cfg, module, agent_num=A, device=DEV)
    hp = po.hyper_from_cfg(cfg)
    pspec, cspec = po.TowerSpec(Dp, K, po.HEAD_CATEGORICAL), po.TowerSpec(Dc, 1, po.HEAD_VALUE)
    ptheta, ctheta = module.models["policy"].theta.cpu().clone(), module.models["critic"].theta.cpu().clone()
    vn = po.ValueNormOracle()
    adv = po.advantages(host["returns"], host["value_preds"], host["active_masks"], vn, False)
    fr = po.flat_rows
    sample = (fr(host["critic_obs"][:-1]), fr(host["policy_obs"][:-1]), fr(host["actions"]),
              fr(host["value_preds"][:-1]), fr(host["returns"][:-1]), fr(host["active_masks"][:-1]),
              fr(host["action_log_probs"]), adv.reshape(-1, 1), np.ones((T * N * A, K), np.float32))
    info_o, gp, gc = po.ppo_update(hp, pspec, ptheta, cspec, ctheta, po.AdamOracle(ptheta.numel(), cfg.lr),
                                   po.AdamOracle(ctheta.numel(), cfg.critic_lr), vn, sample)
    algo._advantages_and_records(buf)
    algo._info.zero_()
    algo._update_minibatch(buf, None, adv.size, True)
    got_p, got_c = module.models["policy"].grad.cpu().numpy(), module.models["critic"].grad.cpu().numpy()
    np.testing.assert_allclose(got_p, gp, rtol=2e-3, atol=3e-5 * np.abs(gp).max() + 1e-7)
    np.testing.assert_allclose(got_c, gc, rtol=2e-3, atol=3e-5 * np.abs(gc).max() + 1e-7)
    want = np.array([info_o[k] for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm",
                                         "critic_grad_norm", "ratio")])
    np.testing.assert_allclose(algo._info[:6].cpu().numpy(), want, rtol=3e-4, atol=3e-5)

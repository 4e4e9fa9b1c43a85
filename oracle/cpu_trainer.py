"""CPU port of the reference's collect + PPO-update loop - TEST INFRASTRUCTURE / ``cpu_baseline`` leg.

Follows ``openrl/drivers/rl_driver.py:118-169`` and ``openrl/drivers/onpolicy_driver.py:57-279`` step by
step on host numpy / torch-CPU, including the data movement the reference really performs on this path
(``np.concatenate(data[step])`` per field, ``np.split`` of every network output, the per-step copies of
the two ``[N, A, recurrent_N, hidden]`` rnn-state arrays an MLP policy never reads, the 12-array
fancy-index gather per minibatch) - that traffic is part of what the reference's CPU path costs
 and is why this is an honest "port" baseline rather than an optimised CPU PPO.

Only ``bench.py``'s ``cpu_baseline`` leg and the tests use it.  It shares every numeric definition with
``oracle/ppo_oracle.py`` (which is pinned against the real reference).
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch

from . import ppo_oracle as po


class CPUReplayData:
    """Host buffer with the reference layout (replay_data.py:41-184)."""

    def __init__(self, T, N, A, D, act_shape, K, H=64, recurrent_N=1):
        f = np.float32
        self.T, self.N, self.A = T, N, A
        self.policy_obs = np.zeros((T + 1, N, A, D), f)
        self.critic_obs = np.zeros((T + 1, N, A, D), f)
        self.rnn_states = np.zeros((T + 1, N, A, recurrent_N, H), f)
        self.rnn_states_critic = np.zeros_like(self.rnn_states)
        self.value_preds = np.zeros((T + 1, N, A, 1), f)
        self.returns = np.zeros_like(self.value_preds)
        self.action_masks = np.ones((T + 1, N, A, K), f) if K else None
        self.actions = np.zeros((T, N, A, act_shape), f)
        self.action_log_probs = np.zeros((T, N, A, act_shape), f)
        self.rewards = np.zeros((T, N, A, 1), f)
        self.masks = np.ones((T + 1, N, A, 1), f)
        self.bad_masks = np.ones_like(self.masks)
        self.active_masks = np.ones_like(self.masks)
        self.step = 0

    def get_batch_data(self, name, step):  # replay_data.py:186-199
        d = getattr(self, name)
        return None if d is None else np.concatenate(d[step])

    def insert(self, obs, rnn, rnn_c, actions, logp, values, rewards, masks, bad, active, amask=None):  # :245-284
        s = self.step
        self.critic_obs[s + 1] = obs.copy()
        self.policy_obs[s + 1] = obs.copy()
        self.rnn_states[s + 1] = rnn.copy()
        self.rnn_states_critic[s + 1] = rnn_c.copy()
        self.actions[s] = actions.copy()
        self.action_log_probs[s] = logp.copy()
        self.value_preds[s] = values.copy()
        self.rewards[s] = rewards.copy()
        self.masks[s + 1] = masks.copy()
        self.bad_masks[s + 1] = bad.copy()
        self.active_masks[s + 1] = active.copy()
        if amask is not None:
            self.action_masks[s + 1] = amask.copy()
        self.step = (s + 1) % self.T

    def after_update(self):  # :300-318
        for f in ("critic_obs", "policy_obs", "rnn_states", "rnn_states_critic", "masks", "bad_masks", "active_masks"):
            a = getattr(self, f)
            a[0] = a[-1].copy()
        if self.action_masks is not None:
            self.action_masks[0] = self.action_masks[-1].copy()


class CPUTrainer:
    def __init__(self, n_envs: int, T: int, obs_dim: int = 4, n_actions: int = 2, seed: int = 0, ppo_epoch: int = 10,
                 num_mini_batch: int = 1, episode_limit: int = 200, hp: po.PPOHyper = None, lr: float = 5e-4,
                 threads: int = None, env=None):
        if threads:
            torch.set_num_threads(threads)
        self.N, self.T, self.D, self.K = n_envs, T, obs_dim, n_actions
        self.hp = hp or po.PPOHyper()
        self.ppo_epoch, self.nmb = ppo_epoch, num_mini_batch
        import random

        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        self.pspec = po.TowerSpec(obs_dim, n_actions, po.HEAD_CATEGORICAL)
        self.cspec = po.TowerSpec(obs_dim, 1, po.HEAD_VALUE)
        self.ptheta = po.init_tower(self.pspec, 0.01)
        self.ctheta = po.init_tower(self.cspec, 1.0)
        self.padam = po.AdamOracle(self.ptheta.numel(), lr)
        self.cadam = po.AdamOracle(self.ctheta.numel(), lr)
        self.vn = po.ValueNormOracle() if self.hp.use_valuenorm else None
        # env: any duck-typed host env (reset() -> [N, 1, D], step(actions) -> obs, rewards, dones, infos), e.g.
        # po.CartPoleEnvOracle for the learning comparison of tests/test_learning_gpu.py
        self.env = env if env is not None else po.SynthEnvOracle(n_envs, obs_dim, seed, episode_limit)
        self.buf = CPUReplayData(T, n_envs, 1, obs_dim, 1, n_actions)
        self.buf.policy_obs[0] = self.env.reset()
        self.buf.critic_obs[0] = self.buf.policy_obs[0]
        self.phase = {"act": 0.0, "env": 0.0, "insert": 0.0, "gae": 0.0, "update": 0.0}

    @torch.no_grad()
    def _act(self, step):  # onpolicy_driver.py:235-279
        b, N = self.buf, self.N
        cobs, pobs = b.get_batch_data("critic_obs", step), b.get_batch_data("policy_obs", step)
        rnn, rnn_c = b.get_batch_data("rnn_states", step), b.get_batch_data("rnn_states_critic", step)
        b.get_batch_data("masks", step)
        am = b.get_batch_data("action_masks", step)
        logits = po.tower_forward(self.pspec, self.ptheta, torch.from_numpy(pobs))
        logits = po.masked_logits(logits, torch.from_numpy(am))
        dist = torch.distributions.Categorical(logits=logits)
        action = dist.sample().unsqueeze(-1)  # FixedCategorical.sample (distributions.py:17-18) -> torch.multinomial
        logp = dist.log_prob(action.squeeze(-1)).view(action.size(0), -1).sum(-1).unsqueeze(-1)
        value = po.tower_forward(self.cspec, self.ctheta, torch.from_numpy(cobs))
        sp = lambda x: np.array(np.split(x, N))
        return (sp(value.numpy()), sp(action.float().numpy()), sp(logp.numpy()), sp(rnn), sp(rnn_c))

    def rollout(self):
        b, N = self.buf, self.N
        for step in range(self.T):
            t0 = time.perf_counter()
            values, actions, logp, rnn, rnn_c = self._act(step)
            t1 = time.perf_counter()
            obs, rewards, dones, infos = self.env.step(actions)
            t2 = time.perf_counter()
            # add2buffer (onpolicy_driver.py:80-152)
            dones_env = np.all(dones, axis=1)
            rnn[dones_env] = 0.0
            rnn_c[dones_env] = 0.0
            masks = np.ones((N, 1, 1), np.float32)
            masks[dones_env] = 0.0
            active = np.ones((N, 1, 1), np.float32)
            active[dones] = 0.0
            active[dones_env] = 1.0
            bad = np.array([[[0.0] if "bad_transition" in info and info["bad_transition"][0] else [1.0]] for info in infos])
            b.insert(obs, rnn, rnn_c, actions, logp, values, rewards, masks, bad, active)
            t3 = time.perf_counter()
            self.phase["act"] += t1 - t0
            self.phase["env"] += t2 - t1
            self.phase["insert"] += t3 - t2

    def update(self) -> Dict[str, float]:
        b = self.buf
        t0 = time.perf_counter()
        with torch.no_grad():  # compute_returns (onpolicy_driver.py:205-233)
            nv = po.tower_forward(self.cspec, self.ctheta, torch.from_numpy(b.get_batch_data("critic_obs", -1)))
        next_values = np.array(np.split(nv.numpy(), self.N))
        vn = self.vn if self.hp.use_valuenorm else None
        b.returns, b.value_preds = po.compute_returns(b.rewards, b.value_preds, b.masks, b.bad_masks, next_values, 0.99,
                                                      0.95, True, False, vn)
        t1 = time.perf_counter()
        bufd = dict(critic_obs=b.critic_obs, policy_obs=b.policy_obs, actions=b.actions, value_preds=b.value_preds,
                    returns=b.returns, active_masks=b.active_masks, action_log_probs=b.action_log_probs,
                    action_masks=b.action_masks)

        def index_fn(M, nmb):  # the reference also gathers rnn_states / masks for every minibatch (:627-636)
            batches = po.feed_forward_indices(M, nmb)
            rs = b.rnn_states[:-1].reshape(-1, *b.rnn_states.shape[3:])
            rc = b.rnn_states_critic[:-1].reshape(-1, *b.rnn_states_critic.shape[3:])
            mk = b.masks[:-1].reshape(-1, 1)
            for idx in batches:
                rs[idx]; rc[idx]; mk[idx]
            return batches

        info, _, _ = po.train_ppo(self.hp, self.pspec, self.ptheta, self.cspec, self.ctheta, self.padam, self.cadam, vn,
                                  bufd, self.ppo_epoch, self.nmb, index_fn=index_fn)
        b.after_update()
        t2 = time.perf_counter()
        self.phase["gae"] += t1 - t0
        self.phase["update"] += t2 - t1
        return info

    def iterate(self) -> Dict[str, float]:
        self.rollout()
        return self.update()


def time_cpu_baseline_bounded(n_envs=4096, ppo_epoch=10, target_seconds=15.0, max_threads=16) -> Dict[str, float]:
    """Bounded CPU sample for bench.py: a short probe (T=8) sizes the rollout length T in {8,...,128} so that
    one iteration (T-step rollout of ``n_envs`` + ``ppo_epoch`` full-batch epochs) takes about
    ``target_seconds``.  Threads are capped: with hundreds of host cores torch's intra-op pool makes these
    small ops SLOWER (measured 900 env-steps/s at 256 threads vs 2.9e4 at 8)."""
    import os

    threads = max(1, min(os.cpu_count() or 1, max_threads))
    probe = time_cpu_baseline(n_envs=n_envs, T=8, ppo_epoch=ppo_epoch, iters=1, threads=threads)
    T = 8
    while T < 128 and probe["seconds"] * (2 * T / 8.0) <= target_seconds:
        T *= 2
    res = probe if T == 8 else time_cpu_baseline(n_envs=n_envs, T=T, ppo_epoch=ppo_epoch, iters=1, threads=threads)
    res["T"] = T
    return res


def time_cpu_baseline(n_envs=4096, T=16, ppo_epoch=10, iters=1, warmup=0, threads=None) -> Dict[str, float]:
    """Bounded sample of the cfg-2 workload: ``iters`` iterations of (T-step rollout of n_envs + ppo_epoch
    full-batch epochs).  Returns env-steps/s and the per-phase split."""
    import os

    threads = threads or os.cpu_count()
    tr = CPUTrainer(n_envs, T, ppo_epoch=ppo_epoch, threads=threads)
    for _ in range(warmup):
        tr.iterate()
    tr.phase = {k: 0.0 for k in tr.phase}
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.iterate()
    dt = time.perf_counter() - t0
    out = {"env_steps_per_s": n_envs * T * iters / dt, "seconds": dt, "cores": threads}
    out.update({"phase_" + k: v for k, v in tr.phase.items()})
    return out

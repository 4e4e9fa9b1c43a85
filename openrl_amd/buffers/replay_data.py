"""Device-resident ``ReplayData`` - same field names, shapes and index contracts as the reference's
host-numpy buffer (``openrl/buffers/replay_data.py:40-184``), stored as
contiguous float32 HIP tensors ``[T(+1), N, A, width]`` in HBM so the rollout, GAE and update kernels
work on it in place and ``buffer.data.<field>`` stays inspectable.

Differences that are deliberate:
* ``rnn_states`` / ``rnn_states_critic`` are zero-stride views unless a recurrent policy is configured
  (the reference allocates and copies 2 x 135 MB of them per rollout at the 4096-env shape for an MLP
  policy that never reads them);
* when the observation space is not a ``Dict{"policy","critic"}``, ``critic_obs`` aliases ``policy_obs``;
* ``records`` / ``advantages`` are engine-side scratch for the fused update (one 64-byte row per sample
  at the CartPole shape).
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .. import ops, spaces


class ReplayData(object):
    def __init__(self, cfg, num_agents, obs_space, act_space, data_client=None, episode_length=None, device=None):
        if episode_length is None:
            episode_length = cfg.episode_length
        self.episode_length = T = int(episode_length)
        self.n_rollout_threads = N = int(cfg.n_rollout_threads)
        self.num_agents = A = int(num_agents)
        self.hidden_size = cfg.rnn_hidden_size if hasattr(cfg, "rnn_hidden_size") else cfg.hidden_size
        self.recurrent_N = cfg.recurrent_N
        self.gamma = cfg.gamma
        self.gae_lambda = cfg.gae_lambda
        self._use_gae = cfg.use_gae
        self._use_popart = cfg.use_popart
        self._use_valuenorm = cfg.use_valuenorm
        self._use_proper_time_limits = cfg.use_proper_time_limits
        self._mixed_obs = False
        if device is None:
            device = getattr(cfg, "device", "cuda:0")
        self.device = dev = nat.require_gpu(device)

        p_space, c_space = spaces.policy_obs_space(obs_space), spaces.critic_obs_space(obs_space)
        self.Dp, self.Dc = spaces.obs_dim(p_space), spaces.obs_dim(c_space)
        self._split_obs = spaces.kind(obs_space) == "Dict"
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        o = lambda *s: torch.ones(*s, dtype=torch.float32, device=dev)
        self.policy_obs = z(T + 1, N, A, self.Dp)
        self.critic_obs = z(T + 1, N, A, self.Dc) if self._split_obs else self.policy_obs
        recurrent = bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy)
        if recurrent:
            self.rnn_states = z(T + 1, N, A, self.recurrent_N, self.hidden_size)
            self.rnn_states_critic = z(T + 1, N, A, self.recurrent_N, self.hidden_size)
        else:
            self.rnn_states = torch.zeros(1, device=dev).expand(T + 1, N, A, self.recurrent_N, self.hidden_size)
            self.rnn_states_critic = self.rnn_states
        self.value_preds = z(T + 1, N, A, 1)
        self.returns = z(T + 1, N, A, 1)
        self.act_kind = spaces.kind(act_space)
        #: actions are integer indices stored as floats (what a host env must be handed as int64)
        self.act_is_index = self.act_kind in ("Discrete", "MultiDiscrete", "MultiBinary")
        self.K = int(act_space.n) if self.act_kind == "Discrete" else 0
        self.action_masks = o(T + 1, N, A, self.K) if self.K else None
        self.act_shape = a = spaces.act_shape(act_space)
        self.actions = z(T, N, A, a)
        self.action_log_probs = z(T, N, A, a)
        self.rewards = z(T, N, A, 1)
        self.masks = o(T + 1, N, A, 1)
        self.bad_masks = o(T + 1, N, A, 1)
        self.active_masks = o(T + 1, N, A, 1)
        self.step = 0

        # engine-side scratch
        self.advantages = torch.empty(T, N, A, 1, dtype=torch.float32, device=dev)
        self.stat_partials = torch.zeros(ops.gae_max_partials(T, N * A), 8, dtype=torch.float64, device=dev)
        self.n_partials = 0
        self._adv_fresh = False  # advantages/stat_partials match returns/value_preds
        self.record_width = ops.record_width(self.Dp, self.Dc, a, self.K)
        self.records: Optional[torch.Tensor] = None  # allocated on first update

    # ------------------------------------------------------------------ reference API
    def get_batch_data(self, data_name: str, step: int):
        """``np.concatenate(data[step])`` -> rows n*A+a (replay_data.py:186-199); a view, no copy."""
        assert hasattr(self, data_name)
        data = getattr(self, data_name)
        if data is None:
            return None
        x = data[step]
        return x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])

    def _dev(self, x, like: torch.Tensor) -> torch.Tensor:
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device, non_blocking=True)
        return t.reshape(like.shape)

    def _obs_pair(self, raw_obs):
        if isinstance(raw_obs, dict):
            return raw_obs.get("policy", raw_obs), raw_obs.get("critic", raw_obs)
        return raw_obs, raw_obs

    def insert(self, raw_obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards, masks,
               bad_masks=None, active_masks=None, action_masks=None):
        """Slot semantics of replay_data.py:245-284 (obs/masks -> step+1, the rest -> step)."""
        s = self.step
        p_obs, c_obs = self._obs_pair(raw_obs)
        self.policy_obs[s + 1].copy_(self._dev(p_obs, self.policy_obs[0]))
        if self.critic_obs is not self.policy_obs:
            self.critic_obs[s + 1].copy_(self._dev(c_obs, self.critic_obs[0]))
        if rnn_states is not None and self.rnn_states.stride(0) != 0:
            self.rnn_states[s + 1].copy_(self._dev(rnn_states, self.rnn_states[0]))
        if rnn_states_critic is not None and self.rnn_states_critic.stride(0) != 0:
            self.rnn_states_critic[s + 1].copy_(self._dev(rnn_states_critic, self.rnn_states_critic[0]))
        self.actions[s].copy_(self._dev(actions, self.actions[0]))
        self.action_log_probs[s].copy_(self._dev(action_log_probs, self.action_log_probs[0]))
        self.value_preds[s].copy_(self._dev(value_preds, self.value_preds[0]))
        self.rewards[s].copy_(self._dev(rewards, self.rewards[0]))
        self.masks[s + 1].copy_(self._dev(masks, self.masks[0]))
        if bad_masks is not None:
            self.bad_masks[s + 1].copy_(self._dev(bad_masks, self.bad_masks[0]))
        if active_masks is not None:
            self.active_masks[s + 1].copy_(self._dev(active_masks, self.active_masks[0]))
        if action_masks is not None and self.action_masks is not None:
            self.action_masks[s + 1].copy_(self._dev(action_masks, self.action_masks[0]))
        self.step = (self.step + 1) % self.episode_length
        self._adv_fresh = False

    def init_buffer(self, raw_obs, action_masks=None):
        p_obs, c_obs = self._obs_pair(raw_obs)
        self.policy_obs[0].copy_(self._dev(p_obs, self.policy_obs[0]))
        if self.critic_obs is not self.policy_obs:
            self.critic_obs[0].copy_(self._dev(c_obs, self.critic_obs[0]))
        if action_masks is not None and self.action_masks is not None:
            self.action_masks[0].copy_(self._dev(action_masks, self.action_masks[0]))

    def after_update(self):
        assert self.step == 0, "step:{} episode:{}".format(self.step, self.episode_length)
        pairs = [(self.policy_obs[0], self.policy_obs[-1])]
        if self.critic_obs is not self.policy_obs:
            pairs.append((self.critic_obs[0], self.critic_obs[-1]))
        if self.rnn_states.stride(0) != 0:
            pairs += [(self.rnn_states[0], self.rnn_states[-1]), (self.rnn_states_critic[0], self.rnn_states_critic[-1])]
        pairs += [(self.masks[0], self.masks[-1]), (self.bad_masks[0], self.bad_masks[-1]),
                  (self.active_masks[0], self.active_masks[-1])]
        if self.action_masks is not None:
            pairs.append((self.action_masks[0], self.action_masks[-1]))
        ops.multi_copy(pairs)  # one launch (orl_multi_copy) instead of one copy per array

    def compute_returns(self, next_value, value_normalizer=None):
        """K6 on the device (replay_data.py:320-423) + the fused advantage statistics (K7a)."""
        nv = self._dev(next_value, self.value_preds[0])
        vn_state = None
        if (self._use_popart or self._use_valuenorm) and value_normalizer is not None:
            vn_state = value_normalizer.state
        self.n_partials = ops.gae_scan(self.rewards, self.value_preds, self.masks,
                                       self.bad_masks if self._use_proper_time_limits else None, nv, vn_state,
                                       self.returns, self.gamma, self.gae_lambda, self._use_gae,
                                       self._use_proper_time_limits, active_masks=self.active_masks,
                                       adv_raw=self.advantages, stat_partials=self.stat_partials)
        self._adv_fresh = True
        self._adv_vn = vn_state

    # ------------------------------------------------------------------ generators (inspection / parity API)
    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None,
                               critic_obs_process_func=None) -> Iterator[Tuple]:
        """Same 12-tuples, same row order and the SAME permutation stream as replay_data.py:553-646:
        ``torch.randperm`` on the host CPU generator (== BatchSampler(SubsetRandomSampler)), gathered on
        the device by one K8 launch per minibatch."""
        T, N, A = self.rewards.shape[0:3]
        batch_size = N * T * A
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch
            mini_batch_size = batch_size // num_mini_batch
        rand = torch.randperm(batch_size)
        n_batches = batch_size // mini_batch_size  # drop_last=True
        flat = lambda x: x.reshape(-1, x.shape[-1])
        rnn_w = self.recurrent_N * self.hidden_size
        srcs = [flat(self.critic_obs[:-1]), flat(self.policy_obs[:-1]), flat(self.actions), flat(self.value_preds[:-1]),
                flat(self.returns[:-1]), flat(self.masks[:-1]), flat(self.active_masks[:-1]),
                flat(self.action_log_probs)]
        if advantages is not None:
            srcs.append(self._dev(advantages, self.advantages).reshape(-1, 1))
        if self.action_masks is not None:
            srcs.append(flat(self.action_masks[:-1]))
        for b in range(n_batches):
            idx = rand[b * mini_batch_size:(b + 1) * mini_batch_size].to(self.device)
            out = ops.gather_minibatch([s.contiguous() for s in srcs], idx)
            k = 8
            adv_targ = None
            if advantages is not None:
                adv_targ = out[k]
                k += 1
            amask = out[k] if self.action_masks is not None else None
            rnn = torch.zeros(mini_batch_size, self.recurrent_N, self.hidden_size, device=self.device)
            if self.rnn_states.stride(0) != 0:
                rs = self.rnn_states[:-1].reshape(-1, rnn_w)
                rc = self.rnn_states_critic[:-1].reshape(-1, rnn_w)
                g = ops.gather_minibatch([rs, rc], idx)
                rnn_a, rnn_c = g[0].view(-1, self.recurrent_N, self.hidden_size), g[1].view(-1, self.recurrent_N,
                                                                                            self.hidden_size)
            else:
                rnn_a = rnn_c = rnn
            critic_obs_batch = out[0]
            if critic_obs_process_func is not None:
                critic_obs_batch = critic_obs_process_func(critic_obs_batch)
            yield (critic_obs_batch, out[1], rnn_a, rnn_c, out[2], out[3], out[4], out[5], out[6], out[7], adv_targ,
                   amask)

    # ------------------------------------------------------------------ engine helpers
    def buffer_ptrs(self) -> nat.BufferPtrs:
        f = nat.fptr
        return nat.BufferPtrs(f(self.policy_obs), f(self.critic_obs), f(self.rewards), f(self.masks), f(self.bad_masks),
                              f(self.active_masks), f(self.action_masks), self.episode_length, self.n_rollout_threads,
                              self.num_agents, self.Dp, self.Dc, self.K)

    def pack_src(self) -> nat.PackSrc:
        f = nat.fptr
        return nat.PackSrc(f(self.policy_obs), f(self.critic_obs), f(self.actions), f(self.action_log_probs),
                           f(self.value_preds), f(self.returns), f(self.active_masks), f(self.action_masks), self.Dp,
                           self.Dc, self.act_shape, self.K)

    def ensure_records(self) -> torch.Tensor:
        if self.records is None:
            M = self.episode_length * self.n_rollout_threads * self.num_agents
            self.records = torch.empty(M, self.record_width, dtype=torch.float32, device=self.device)
        return self.records

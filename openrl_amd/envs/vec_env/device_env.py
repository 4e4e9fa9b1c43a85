"""Device-resident batched environments behind the VecEnv duck-type the on-policy driver consumes
(``openrl/envs/vec_env/base_venv.py:38-358``; precedent for a batched adaptor:
``examples/isaac/isaac2openrl.py:28-88``).

``parallel_env_num, agent_num, observation_space, action_space, env_name, use_monitor,
reset(seed=...) -> (obs, info), step(actions, extra_data) -> (obs, rewards, dones, infos),
batch_rewards(buffer), statistics(buffer), close()``.

State lives in HBM; ``step`` launches ``orl_env_step``; the fused driver never calls ``step`` at all -
it hands ``device_handle()`` to ``orl_rollout_fused`` which advances the same state in-kernel.
Two kinds are built:

* ``synthetic``: the fixed-step benchmark env - obs ~ N(0,1) keyed (seed, env, t), reward U(0,1),
  episodes of exactly ``episode_limit`` steps with per-env phase (env*7) mod limit, no bad transitions;
* ``cartpole``: CartPole-v1 dynamics (gymnasium classic_control, euler integrator, fp32), 500-step limit,
  auto-reset, ``done = terminated or truncated`` (RemoveTruncated, envs/wrappers/extra_wrappers.py:122-134).

Env ``i`` is seeded ``seed + i*10086`` in the reference (sync_venv.py:136-137); here the counter-based
generator is keyed by ``(seed, i)`` which gives every lane its own stream the same way.
"""
from __future__ import annotations

import time
from typing import Any, Dict, Optional

import numpy as np
import torch

from ... import _native as nat
from ... import ops, spaces


class DeviceVecEnv:
    def __init__(self, kind: str, env_num: int, obs_dim: int, action_space, env_name: str, episode_limit: int,
                 device="cuda:0", seed: int = 0):
        self.kind = kind
        self.env_kind = {"synthetic": ops.ENV_SYNTH, "cartpole": ops.ENV_CARTPOLE}[kind]
        self.device = nat.require_gpu(device)
        self._n = int(env_num)
        self._obs_dim = int(obs_dim)
        self._action_space = action_space
        self._observation_space = spaces.Box(-np.inf, np.inf, (obs_dim,), np.float32)
        self._env_name = env_name
        self.episode_limit = int(episode_limit)
        self.seed = int(seed)
        w = ops.env_state_width(self.env_kind)
        self.env_state = torch.zeros(self._n, w, dtype=torch.float32, device=self.device)
        self.ep_stats = torch.zeros(self._n, 4, dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(self._n, 1, self._obs_dim, dtype=torch.float32, device=self.device)
        self._rew = torch.zeros(self._n, dtype=torch.float32, device=self.device)
        self._done = torch.zeros(self._n, dtype=torch.uint8, device=self.device)
        self.global_step = 0
        self.start_time = time.time()
        self.total_step = 0
        self._infos = [{} for _ in range(self._n)]
        self.is_device_env = True
        self.supports_graph_rollout = True  # stepwise mode: the step counter has a device part (orl_env_step_dev)

    # ---- VecEnv contract
    @property
    def parallel_env_num(self) -> int:
        return self._n

    @property
    def agent_num(self) -> int:
        return 1

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def action_space(self):
        return self._action_space

    @property
    def env_name(self) -> str:
        return self._env_name

    @property
    def use_monitor(self) -> bool:
        return True

    def reset_device(self, seed: Optional[int] = None) -> torch.Tensor:
        if seed is not None:
            self.seed = int(seed)
        ops.env_reset(self.env_kind, self.env_state, self.ep_stats, self.obs, self._n, self._obs_dim, self.seed,
                      self.episode_limit)
        self.global_step = 0
        return self.obs

    def reset(self, seed: Optional[int] = None, options=None):
        obs = self.reset_device(seed)
        return obs.cpu().numpy(), {}

    def step_device(self, actions: Optional[torch.Tensor]):
        a = None
        if actions is not None:
            a = actions.to(self.device, torch.float32).reshape(self._n, -1).contiguous()
        # while the driver captures a stepwise rollout into a hipGraph it hands the env the graph's device step counter
        # (``rng_step_dev``) and that counter's value at capture (``rng_step_host``): host part + device part == the
        # env's own step count on every replay
        rdev = getattr(self, "rng_step_dev", None)
        host = self.global_step - (int(getattr(self, "rng_step_host", 0)) if rdev is not None else 0)
        ops.env_step(self.env_kind, self.env_state, self.ep_stats, a, self.obs, self._rew, self._done, self._n,
                     self._obs_dim, self.seed, self.episode_limit, host, rdev)
        self.global_step += 1
        return self.obs, self._rew.view(self._n, 1, 1), self._done.view(self._n, 1)

    def step(self, actions, extra_data: Optional[Dict[str, Any]] = None):
        a = None if actions is None else torch.as_tensor(np.asarray(actions), dtype=torch.float32)
        obs, rew, done = self.step_device(a)
        return obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), self._infos

    def random_action(self, infos=None):
        """base_venv.py:312-336: one ``action_space.sample`` per (env, agent) -> [N, A, 1] for Discrete, [N, A, dim]
        otherwise; ``infos`` (what ``reset`` / ``step`` returned) supply legal-move masks."""
        from ...drivers.onpolicy_driver import prepare_action_masks

        masks = prepare_action_masks(infos, self.agent_num)
        sp, out = self.action_space, []
        for n in range(self.parallel_env_num):
            row = []
            for a in range(self.agent_num):
                m = None if masks is None else np.asarray(masks[n]).reshape(self.agent_num, -1)[a]
                x = sp.sample(mask=m) if m is not None else sp.sample()
                row.append(np.asarray(x).reshape(-1))
            out.append(row)
        return np.array(out)

    def batch_rewards(self, buffer) -> Dict[str, Any]:
        return {}

    def count_steps(self, buffer) -> None:
        d = buffer.data if hasattr(buffer, "data") else buffer
        self.total_step += d.rewards.shape[0] * d.rewards.shape[1]

    def statistics(self, buffer) -> Dict[str, Any]:
        """SimpleVecInfo.statistics (vec_info/simple_vec_info.py:18-32): FPS + mean rollout reward."""
        d = buffer.data if hasattr(buffer, "data") else buffer
        self.count_steps(buffer)
        r = d.rewards.mean(dim=1).sum(dim=(0, 2))  # per agent: mean over envs, sum over time
        vals = r.cpu().tolist()
        info = {"agent_%d/rollout_episode_reward" % i: v for i, v in enumerate(vals)}
        info["FPS"] = int(self.total_step / max(time.time() - self.start_time, 1e-9))
        info["rollout_episode_reward"] = float(np.mean(vals))
        return info

    def episode_statistics(self) -> Dict[str, float]:
        """Finished-episode return statistics tracked in-kernel (``ep_stats``)."""
        s = self.ep_stats.sum(dim=0).cpu().tolist()
        n = max(s[3], 1.0)
        return {"episodes_finished": s[3], "episode_return_mean": s[2] / n}

    def close(self):
        pass

    def render(self, *a, **k):
        return None

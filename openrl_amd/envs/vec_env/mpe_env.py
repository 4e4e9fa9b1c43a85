"""Device-resident batched MPE ``simple_spread`` (BASELINE config 4's env) behind the
same VecEnv duck-type as ``device_env.DeviceVecEnv``.

Reference: ``openrl/envs/mpe/mpe_env.py:14-29`` builds ``MultiAgentEnv(world, scenario callbacks)`` per env and
``SyncVectorEnv`` steps 2048 python worlds one by one; here one HIP launch (``orl_mpe_step``) advances every world.
Spaces follow ``multiagent_env.py:86-151``: ``Dict{"policy": Box(18), "critic": Box(54)}``, ``Discrete(5)``,
3 agents, episodes of ``world_length`` = 25 steps with auto-reset."""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from ... import _native as nat
from ... import ops_rnn, spaces
from .device_env import DeviceVecEnv


class MpeSpreadVecEnv(DeviceVecEnv):
    N_AGENTS, OBS, COBS = 3, 18, 54

    def __init__(self, env_num: int, env_name: str = "simple_spread", world_length: int = 25, device="cuda:0",
                 seed: int = 0):
        self.kind = "mpe_simple_spread"
        self.env_kind = nat.ORL_ENV_MPE_SPREAD  # stepped in-kernel by orl_rnn_rollout_fused (recurrent policies)
        self.device = nat.require_gpu(device)
        self._n = int(env_num)
        self._env_name = env_name
        self.episode_limit = int(world_length)
        self.seed = int(seed)
        box = lambda d: spaces.Box(-np.inf, np.inf, (d,), np.float32)
        self._observation_space = spaces.Dict({"policy": box(self.OBS), "critic": box(self.COBS)})
        self._action_space = spaces.Discrete(5)
        z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
        self.env_state = z(self._n, ops_rnn.mpe_state_width())
        self.ep_stats = z(self._n, 4)
        self.obs = {"policy": z(self._n, self.N_AGENTS, self.OBS), "critic": z(self._n, self.N_AGENTS, self.COBS)}
        self._rew = z(self._n, self.N_AGENTS, 1)
        self._done = z(self._n, self.N_AGENTS, dtype=torch.uint8)
        self.global_step = 0
        self.start_time = time.time()
        self.total_step = 0
        self._infos = [{} for _ in range(self._n)]
        self.is_device_env = True
        self.supports_fused_rollout = False     # the MLP rollout kernel steps single-agent envs only
        self.supports_fused_rnn_rollout = True  # recurrent MAPPO: orl_rnn_rollout_fused
        self.supports_graph_rollout = True  # orl_mpe_step takes no per-call host scalar: capturable

    @property
    def agent_num(self) -> int:
        return self.N_AGENTS

    def reset_device(self, seed: Optional[int] = None):
        if seed is not None:
            self.seed = int(seed)
        ops_rnn.mpe_reset(self.env_state, self.ep_stats, self.obs["policy"], self.obs["critic"], self._n, self.seed)
        self.global_step = 0
        return self.obs

    def reset(self, seed: Optional[int] = None, options=None):
        obs = self.reset_device(seed)
        return {k: v.cpu().numpy() for k, v in obs.items()}, {}

    def step_device(self, actions: torch.Tensor):
        a = actions.to(self.device, torch.float32).reshape(self._n, self.N_AGENTS).contiguous()
        ops_rnn.mpe_step(self.env_state, self.ep_stats, a, self.obs["policy"], self.obs["critic"], self._rew,
                         self._done, self._n, self.seed, self.episode_limit)
        self.global_step += 1
        return self.obs, self._rew, self._done

    def step(self, actions, extra_data=None):
        obs, rew, done = self.step_device(torch.as_tensor(np.asarray(actions), dtype=torch.float32))
        return ({k: v.cpu().numpy() for k, v in obs.items()}, rew.cpu().numpy(), done.cpu().numpy().astype(bool),
                self._infos)

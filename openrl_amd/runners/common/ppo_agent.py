"""``PPOAgent`` - the trainer handle of the drop-in API (``openrl/runners/common/ppo_agent.py:39-158`` over
``rl_agent.py:35-216``): same constructor, ``train(total_time_steps, callback, train_algo_class, logger,
driver_class)``, ``act``, ``set_env``, ``save`` / ``load`` / ``load_policy``."""
from __future__ import annotations

import pathlib
import time
from typing import Any, Dict, List, Optional, Tuple, Type, Union

import numpy as np
import torch

from ...algorithms.ppo import PPOAlgorithm
from ...buffers import NormalReplayBuffer as ReplayBuffer
from ...drivers.onpolicy_driver import OnPolicyDriver as Driver
from ...drivers.onpolicy_driver import prepare_action_masks
from ...utils.callbacks import as_callback
from ...utils.logger import Logger
from ...utils.util import _t2n


class PPOAgent:
    def __init__(self, net=None, env=None, run_dir: Optional[str] = None, env_num: Optional[int] = None, rank: int = 0,
                 world_size: int = 1, use_wandb: bool = False, use_tensorboard: bool = False,
                 project_name: str = "PPOAgent") -> None:
        self.net = net
        if self.net is not None:
            self.net.reset()
        self._cfg = net.cfg
        self._use_wandb = use_wandb
        self._use_tensorboard = not use_wandb and use_tensorboard
        self.project_name = project_name
        if env is not None:
            self._env = env
        elif hasattr(net, "env") and net.env is not None:
            self._env = net.env
        else:
            raise ValueError("env is None")
        self.env_num = env_num if env_num is not None else self._env.parallel_env_num
        self.num_time_steps = 0
        self._episode_num = 0
        self._total_time_steps = 0
        self._cfg.n_rollout_threads = self.env_num
        self._cfg.learner_n_rollout_threads = self._cfg.n_rollout_threads
        self.rank = rank
        self.world_size = world_size
        self.client = None
        self.agent_num = self._env.agent_num
        self.run_dir = self._cfg.run_dir if run_dir is None else run_dir
        self.exp_name = "rl" if self._cfg.experiment_name == "" else self._cfg.experiment_name
        self.driver = None

    def train(self, total_time_steps: int, callback=None, train_algo_class: Type = PPOAlgorithm,
              logger: Optional[Logger] = None, driver_class: Type = Driver) -> None:
        self._cfg.num_env_steps = total_time_steps
        self.config = {"cfg": self._cfg, "num_agents": self.agent_num, "run_dir": self.run_dir, "envs": self._env,
                       "device": self.net.device}
        trainer = train_algo_class(cfg=self._cfg, init_module=self.net.module, device=self.net.device,
                                   agent_num=self.agent_num)
        buffer = ReplayBuffer(self._cfg, self.agent_num, self._env.observation_space, self._env.action_space,
                              data_client=None, device=self.net.device)
        if logger is None:
            logger = Logger(cfg=self._cfg, project_name=self.project_name, scenario_name=self._env.env_name,
                            exp_name=self.exp_name, log_path=self.run_dir, use_wandb=self._use_wandb,
                            use_tensorboard=self._use_tensorboard)
        self._logger = logger
        self.start_time = time.time_ns()
        self.num_time_steps = 0
        self._episode_num = 0
        self._total_time_steps = total_time_steps
        callback = as_callback(callback)
        callback.init_callback(self)
        driver = driver_class(config=self.config, trainer=trainer, buffer=buffer, agent=self, client=self.client,
                              rank=self.rank, world_size=self.world_size, logger=logger, callback=callback)
        self.driver = driver
        callback.on_training_start(locals(), globals())
        driver.run()
        callback.on_training_end()

    def act(self, observation: Union[np.ndarray, Dict[str, np.ndarray]], info: Optional[List[Dict[str, Any]]] = None,
            deterministic: bool = True, episode_starts: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Any]:
        assert self.net is not None, "net is None"
        if isinstance(observation, dict):
            observation = observation.get("policy", observation)
        obs = observation if isinstance(observation, torch.Tensor) else np.concatenate(np.asarray(observation), axis=0)
        action_masks = prepare_action_masks(info, self.agent_num) if info is not None else None
        if action_masks is not None:
            action_masks = np.concatenate(action_masks, axis=0)
        action, rnn_state = self.net.act(obs, action_masks=action_masks, deterministic=deterministic,
                                         episode_starts=episode_starts)
        action = np.array(np.split(_t2n(action), self.env_num))
        return action, rnn_state

    def reset(self):
        self.net.reset()

    def get_env(self):
        """base_agent.py:50-57."""
        return self._env

    @property
    def logger(self):
        """base_agent.py:58-61."""
        return getattr(self, "_logger", None)

    def set_env(self, env):
        self.net.reset()
        if env is not None:
            self._env = env
            self.env_num = env.parallel_env_num
            self.agent_num = env.agent_num
        env.reset(seed=self._cfg.seed)
        self.net.reset(env)

    def save(self, path) -> None:
        path = pathlib.Path(path)
        path.mkdir(parents=True, exist_ok=True)
        mod = self.net.module
        torch.save({"models": {k: {n: v.detach().cpu() for n, v in m.state_dict().items()}
                               for k, m in mod.models.items()},
                    "optimizers": {k: {"exp_avg": o.exp_avg.cpu(), "exp_avg_sq": o.exp_avg_sq.cpu(),
                                       "step": o.step_count, "param_groups": o.param_groups}
                                   for k, o in mod.optimizers.items()}}, path / "module.pt")

    def load(self, path) -> None:
        path = pathlib.Path(path)
        assert path.exists(), f"{path} does not exist"
        if path.is_dir():
            path = path / "module.pt"
        blob = torch.load(path, map_location="cpu")
        mod = self.net.module
        for k, m in mod.models.items():
            m.load_state_dict(blob["models"][k])
        for k, o in mod.optimizers.items():
            s = blob["optimizers"][k]
            o.exp_avg.copy_(s["exp_avg"])
            o.exp_avg_sq.copy_(s["exp_avg_sq"])
            o.step_count = int(s["step"])
            o.param_groups = s["param_groups"]
        self.net.reset()

    def load_policy(self, path) -> None:
        self.net.load_policy(path)

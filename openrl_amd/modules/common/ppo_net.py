"""``PPONet(env, cfg, device, n_rollout_threads, model_dict, module_class)`` - the network handle of the
drop-in API (``openrl/modules/common/ppo_net.py:50-143``): seeds, resets the env, completes ``cfg`` and
builds the module with ``rank=0, world_size=1`` semantics (the multi-GPU world size is picked up from
``torch.distributed`` when it is initialised - the reference pins it to 1, ppo_net.py:93-94)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch

from ... import _native as nat
from ... import distributed as dist_utils
from ...configs.config import default_cfg
from ...utils.util import set_seed
from ..ppo_module import PPOModule


class PPONet:
    def __init__(self, env, cfg=None, device: Union[torch.device, str] = "cuda:0", n_rollout_threads: int = 1,
                 model_dict: Optional[Dict[str, Any]] = None, module_class=PPOModule) -> None:
        if cfg is None:
            cfg = default_cfg([])
        set_seed(cfg.seed)
        env.reset(seed=cfg.seed)
        cfg.num_agents = env.agent_num
        cfg.n_rollout_threads = n_rollout_threads
        cfg.learner_n_rollout_threads = cfg.n_rollout_threads
        if cfg.rnn_type == "gru":
            cfg.rnn_hidden_size = cfg.hidden_size
        elif cfg.rnn_type == "lstm":
            cfg.rnn_hidden_size = cfg.hidden_size * 2
        else:
            raise NotImplementedError(f"RNN type {cfg.rnn_type} has not been implemented.")
        device = nat.require_gpu(device)  # no CPU path: the reference's default "cpu" is rejected loudly
        self.module = module_class(cfg=cfg, policy_input_space=env.observation_space,
                                   critic_input_space=env.observation_space, act_space=env.action_space,
                                   share_model=cfg.use_share_model, device=device, rank=dist_utils.rank(),
                                   world_size=dist_utils.world_size(), model_dict=model_dict)
        if dist_utils.world_size() > 1:  # replicas start from rank 0's weights
            for m in self.module.models.values():
                dist_utils.broadcast_(m.theta, 0)
        self.cfg = cfg
        self.env = env
        self.device = device
        self.rnn_states_actor = None
        self.masks = None

    def act(self, observation, action_masks: Optional[np.ndarray] = None, deterministic: bool = False,
            episode_starts: Optional[np.ndarray] = None) -> Tuple[torch.Tensor, Any]:
        if episode_starts is not None and self.rnn_states_actor is not None:
            # reset_rnn_states (ppo_net.py:33-47): zero the state of every agent of an env that starts an episode
            starts = np.repeat(np.asarray(episode_starts, dtype=np.float32).reshape(-1), self.env.agent_num)
            mask = 1.0 - starts[:, None, None]
            if isinstance(self.rnn_states_actor, torch.Tensor):
                self.rnn_states_actor = self.rnn_states_actor * torch.as_tensor(mask, dtype=torch.float32,
                                                                                device=self.rnn_states_actor.device)
            else:
                self.rnn_states_actor = self.rnn_states_actor * mask
        actions, self.rnn_states_actor = self.module.act(obs=observation, rnn_states_actor=self.rnn_states_actor,
                                                         masks=self.masks, action_masks=action_masks,
                                                         deterministic=deterministic)
        return actions, self.rnn_states_actor

    def reset(self, env=None) -> None:
        if env is not None:
            self.env = env
        self.rnn_states_actor, self.masks = self.module.init_rnn_states(
            rollout_num=self.env.parallel_env_num, agent_num=self.env.agent_num, rnn_layers=self.cfg.recurrent_N,
            hidden_size=self.cfg.rnn_hidden_size)

    def load_policy(self, path: str) -> None:
        self.module.load_policy(path)

"""Device-resident batched tic-tac-toe against a uniformly random opponent (BASELINE config 5's env; SURVEY.md
section 8f rank 3) behind the VecEnv duck-type of ``device_env.DeviceVecEnv``.

Reference: ``examples/selfplay`` - ``make("tictactoe_v3", opponent_wrappers=[RandomOpponentWrapper], ...)`` wraps a
PettingZoo game so that the learning agent sees a single-agent env whose step contains the opponent's reply
(``selfplay/wrappers/base_multiplayer_wrapper.py:85-150``), with the legal moves delivered as ``info["action_masks"]``
(``MoveActionMask2InfoWrapper``).  Here one HIP launch (``orl_ttt_step``) advances every game, and the legal-move masks
stay on the device (``action_mask_device``: the driver hands them to ``orl_buffer_insert`` / ``orl_act_step``).
The opponent-pool variant of the reference (``OpponentPoolWrapper``: opponents are earlier checkpoints served by a
selfplay API) is not built."""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from ... import _native as nat
from ... import ops_rnn, spaces
from .device_env import DeviceVecEnv


class TicTacToeVecEnv(DeviceVecEnv):
    OBS, N_ACT = 18, 9

    def __init__(self, env_num: int, env_name: str = "tictactoe_v3", device="cuda:0", seed: int = 0):
        self.kind = "tictactoe_random_opponent"
        self.env_kind = nat.ORL_ENV_TTT  # the fused rollout kernel plays the game in-kernel (orl_rollout_fused)
        self.device = nat.require_gpu(device)
        self._n = int(env_num)
        self._env_name = env_name
        self.episode_limit = 5  # a game lasts at most 5 agent moves
        self.seed = int(seed)
        self._observation_space = spaces.Box(0.0, 1.0, (self.OBS,), np.float32)
        self._action_space = spaces.Discrete(self.N_ACT)
        z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
        self.env_state = z(self._n, ops_rnn.ttt_state_width())
        self.ep_stats = z(self._n, 4)
        self.obs = z(self._n, 1, self.OBS)
        self.action_mask_device = z(self._n, 1, self.N_ACT)  # 1 = legal; read by the driver after reset / step
        self._rew = z(self._n, 1, 1)
        self._done = z(self._n, 1, dtype=torch.uint8)
        self.global_step = 0
        self.start_time = time.time()
        self.total_step = 0
        self._infos = [{} for _ in range(self._n)]
        self.is_device_env = True
        self.supports_fused_rollout = True
        self.supports_graph_rollout = True  # stepwise mode: orl_ttt_step takes no per-call host scalar, capturable

    def reset_device(self, seed: Optional[int] = None):
        if seed is not None:
            self.seed = int(seed)
        ops_rnn.ttt_reset(self.env_state, self.ep_stats, self.obs, self.action_mask_device, self._n, self.seed)
        self.global_step = 0
        return self.obs

    def reset(self, seed: Optional[int] = None, options=None):
        obs = self.reset_device(seed)
        return obs.cpu().numpy(), {"action_masks": self.action_mask_device.cpu().numpy()}

    def step_device(self, actions: torch.Tensor):
        a = actions.to(self.device, torch.float32).reshape(self._n).contiguous()
        ops_rnn.ttt_step(self.env_state, self.ep_stats, a, self.obs, self.action_mask_device, self._rew, self._done,
                         self._n, self.seed)
        self.global_step += 1
        return self.obs, self._rew, self._done

    def step(self, actions, extra_data=None):
        obs, rew, done = self.step_device(torch.as_tensor(np.asarray(actions), dtype=torch.float32))
        masks = self.action_mask_device.cpu().numpy()
        infos = [{"action_masks": masks[n]} for n in range(self._n)]
        return obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), infos


class TicTacToeSelfPlayVecEnv(TicTacToeVecEnv):
    """Self-play variant: the opponent is a POLICY - one of ``pool_size`` frozen snapshots of the learner
    (``openrl/selfplay/wrappers/opponent_pool_wrapper.py:37-120`` plays earlier checkpoints).

    Env group g (the g-th contiguous block of N / pool_size envs) plays snapshot g.  In stepwise mode a step is three device stages
    (the fused rollout kernel does the same in-kernel, ``ORL_ENV_TTT_POOL``):
    ``orl_ttt_agent_move`` -> ONE ``orl_act_step_grouped`` launch that evaluates every snapshot on its group's
    opponent-side boards (sampled under the legal-move masks) -> ``orl_ttt_opponent_move``; all of them capturable, so the stepwise rollout still replays as one
    hipGraph.  Snapshots are refreshed in place with ``push_opponent(theta)`` (round-robin), e.g. by
    ``SelfPlayCallback``.  Until the first push every slot holds all-zero parameters = the uniformly random opponent.

    Which snapshot an env plays (``opponent_sampling``; the reference asks its self-play service for an opponent at
    EVERY ``reset``, opponent_pool_wrapper.py:37-66, and the service applies a sample strategy):

    * ``"per_reset"`` (default, the reference's ``OpponentPoolWrapper.reset``): a finished game draws a fresh slot for
      its env from a Philox stream on the device (``orl_opponent_sample``).  Stepwise, the act launch takes the per-env
      slot (``orl_act_step_pool``); the fused one-launch rollout keeps all (up to 4) snapshot images in LDS, walks the
      opponent tower once per distinct slot a 16-env tile holds and makes the same draws in-kernel
      (``orl_rollout_args.opp_per_reset``); larger pools roll out stepwise / as a hipGraph.
    * ``"per_rollout"``: one draw per 16-env tile at every rollout start - the same marginal distribution of
      opponents, coarser in time, and the fused one-launch rollout still applies (a tile shares one opponent image).
    * ``"static"``: env group g always plays slot g (round 1's behaviour)."""

    STRATEGIES = {"RandomOpponent": 0, "LastOpponent": 1}

    def __init__(self, env_num: int, env_name: str = "tictactoe_v3", device="cuda:0", seed: int = 0, pool_size: int = 4,
                 opponent_sampling: str = "per_reset", opponent_strategy: str = "RandomOpponent"):
        super().__init__(env_num, env_name, device=device, seed=seed)
        from ... import ops

        if opponent_sampling not in ("per_reset", "per_rollout", "static"):
            raise ValueError("opponent_sampling must be per_reset / per_rollout / static, got %r" % opponent_sampling)
        if opponent_strategy not in self.STRATEGIES:
            raise ValueError("opponent_strategy must be one of %s" % sorted(self.STRATEGIES))
        self.opponent_sampling, self.opponent_strategy = opponent_sampling, opponent_strategy
        self.kind = "tictactoe_selfplay_pool"
        self.env_kind = nat.ORL_ENV_TTT_POOL  # orl_rollout_fused plays both sides in-kernel (fill_rollout_args)
        # per_reset inside the fused kernel keeps every snapshot image in LDS next to the learner's: up to 4 fit
        self.supports_fused_rollout = opponent_sampling != "per_reset" or int(pool_size) <= 4
        self.opp_index = torch.zeros(self._n, dtype=torch.int32, device=self.device)
        self._draws = 0
        self.pool_size = max(1, min(int(pool_size), self._n))
        self.opp_net = nat.NetDesc(self.OBS, 64, self.N_ACT, nat.ORL_HEAD_CATEGORICAL)
        self.opp_thetas = torch.zeros(self.pool_size, ops.param_count(self.opp_net), device=self.device)
        self.opp_seed = (self.seed * 2654435761 + 0x5E1F) & (2 ** 63 - 1)
        self.pushes = 0
        z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
        self._opp_obs, self._opp_mask = z(self._n, self.OBS), z(self._n, self.N_ACT)
        self._opp_act, self._opp_lp = z(self._n, 1), z(self._n, 1)
        # contiguous env groups of a multiple of 16 rows (one MFMA tile never mixes two snapshots)
        self._group_rows = -(-self._n // self.pool_size)
        self._group_rows = -(-self._group_rows // 16) * 16
        self.pool_size = -(-self._n // self._group_rows)  # groups that actually receive envs

    def fill_rollout_args(self, args) -> None:
        """The opponent-pool fields of orl_rollout_args (fused rollout); same Philox keys as the stepwise path."""
        args.opp_thetas = self.opp_thetas.data_ptr()
        args.opp_theta_stride = self.opp_thetas.stride(0)
        args.opp_group_rows = self._group_rows
        args.opp_seed = self.opp_seed & (2 ** 64 - 1)
        args.opp_rng_step0 = self.global_step
        if self.opponent_sampling == "per_rollout":
            args.opp_index = self.opp_index.data_ptr()
        elif self.opponent_sampling == "per_reset":  # per-env slots, re-drawn in-kernel when a game ends
            args.opp_index = self.opp_index.data_ptr()
            args.opp_per_reset = 1
            args.opp_n_policies = self.opp_thetas.shape[0]
            args.opp_n_filled = self.n_filled
            args.opp_last_slot = (self.pushes - 1) % self.pool_size if self.pushes else 0
            args.opp_strategy = self.STRATEGIES[self.opponent_strategy]
            args.opp_sample_seed = (self.opp_seed ^ 0x0B0E) & (2 ** 64 - 1)
            args.opp_draw_id0 = self._draws

    def after_fused_rollout(self, steps: int) -> None:
        """Driver hook: the fused kernel consumed one opponent draw id per step (like ``steps`` stepwise steps)."""
        if self.opponent_sampling == "per_reset":
            self._draws += steps

    @property
    def n_filled(self) -> int:
        return max(1, min(self.pushes, self.pool_size))

    def sample_opponents(self, dones, per_tile: bool = False) -> None:
        """Draw a pool slot for every env in ``dones`` (None: all) with the configured strategy - on the device."""
        from ... import ops

        last = (self.pushes - 1) % self.pool_size if self.pushes else 0
        ops.opponent_sample(self.opp_index, dones, self.n_filled, last, self.STRATEGIES[self.opponent_strategy], per_tile,
                            self.opp_seed ^ 0x0B0E, self._draws, getattr(self, "rng_step_dev", None))
        self._draws += 1

    def reset_device(self, seed=None):
        obs = super().reset_device(seed)
        if self.opponent_sampling == "per_reset":
            self.sample_opponents(None)
        return obs

    def on_rollout_start(self) -> None:
        """Driver hook (fused and stepwise rollouts alike): the per-rollout draw of one pool slot per 16-env tile."""
        if self.opponent_sampling == "per_rollout":
            self.sample_opponents(None, per_tile=True)

    def push_opponent(self, theta: torch.Tensor) -> int:
        """Store a snapshot of the learner's policy parameters in the next pool slot (in place: graph-safe)."""
        slot = self.pushes % self.pool_size
        self.opp_thetas[slot].copy_(theta.detach().to(self.device).reshape(-1))
        self.pushes += 1
        return slot

    def step_device(self, actions: torch.Tensor):
        from ... import ops

        a = actions.to(self.device, torch.float32).reshape(self._n).contiguous()
        ops_rnn.ttt_agent_move(self.env_state, a, self._opp_obs, self._opp_mask, self._rew, self._done, self._n)
        rdev = getattr(self, "rng_step_dev", None)
        if self.opponent_sampling == "static":
            ops.act_step_grouped(self.opp_net, self.opp_thetas, self._group_rows, self._opp_obs, self._opp_mask, self._n,
                                 False, self.opp_seed, 0, self.global_step, self._opp_act, self._opp_lp, rng_step_dev=rdev)
        else:  # every env (per_reset) / tile (per_rollout) names its own pool slot
            ops.act_step_pool(self.opp_net, self.opp_thetas, self.opp_index, self._opp_obs, self._opp_mask, self._n, False,
                              self.opp_seed, 0, self.global_step, self._opp_act, self._opp_lp, rng_step_dev=rdev)
        ops_rnn.ttt_opponent_move(self.env_state, self.ep_stats, self._opp_act, self.obs, self.action_mask_device,
                                  self._rew, self._done, self._n, self.seed)
        if self.opponent_sampling == "per_reset":  # a finished game -> a fresh opponent for the next one
            self.sample_opponents(self._done)
        self.global_step += 1
        return self.obs, self._rew, self._done

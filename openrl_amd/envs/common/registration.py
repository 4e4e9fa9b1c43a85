"""``make(id, env_num, ...)`` - the env factory of the drop-in API
(``openrl/envs/common/registration.py:35-182``).

The reference's dispatch table builds per-env python gym objects inside ``SyncVectorEnv`` /
``AsyncVectorEnv`` (sync_venv.py:63-64, async_venv.py:45-875) - the CPU env path, which this engine does
not rebuild.  What IS built are the device-resident batched envs of ``vec_env/device_env.py``; any other
env comes in through ``make_custom_envs`` (same hook as registration.py:64-67) as a duck-typed VecEnv.
"""
from __future__ import annotations

from typing import Callable, Optional

from ... import spaces
from ..vec_env.device_env import DeviceVecEnv

_SYNTH_PREFIX = "SyntheticFixedStep"


def make(id: str, env_num: int = 1, asynchronous: bool = False, add_monitor: bool = True,
         render_mode: Optional[str] = None, make_custom_envs: Optional[Callable] = None, auto_reset: bool = True,
         **kwargs):
    """Same signature as the reference.  Extra keyword arguments understood here: ``device`` (default
    "cuda:0"), ``seed``, and for the synthetic env ``obs_dim``, ``action_space``, ``episode_limit``."""
    device = kwargs.pop("device", "cuda:0")
    seed = kwargs.pop("seed", 0)
    if make_custom_envs is not None:
        return make_custom_envs(id=id, env_num=env_num, render_mode=render_mode, **kwargs)
    if id == "CartPole-v1":
        return DeviceVecEnv("cartpole", env_num, 4, spaces.Discrete(2), id, kwargs.pop("episode_limit", 500),
                            device=device, seed=seed)
    if id.startswith(_SYNTH_PREFIX):
        obs_dim = kwargs.pop("obs_dim", 4)
        act = kwargs.pop("action_space", spaces.Discrete(2))
        return DeviceVecEnv("synthetic", env_num, obs_dim, act, id, kwargs.pop("episode_limit", 200), device=device,
                            seed=seed)
    if id == "simple_spread":  # openrl/envs/mpe (registration.py:76-81 -> make_mpe_envs)
        from ..vec_env.mpe_env import MpeSpreadVecEnv

        return MpeSpreadVecEnv(env_num, id, kwargs.pop("world_length", 25), device=device, seed=seed)
    if id == "tictactoe_v3":  # openrl/envs/PettingZoo + selfplay RandomOpponentWrapper (examples/selfplay)
        from ..vec_env.tictactoe_env import TicTacToeSelfPlayVecEnv, TicTacToeVecEnv

        opponent = kwargs.pop("opponent", "random")
        # the reference selects the opponent with wrapper classes (examples/selfplay/train_selfplay.py:26,51:
        # opponent_wrappers=[RecordWinner, OpponentPoolWrapper] / [..., RandomOpponentWrapper]); their NAMES (classes or
        # strings) are honoured here, the wrappers' work itself (opponent moves, winner records) is in-kernel
        names = [w if isinstance(w, str) else getattr(w, "__name__", str(w)) for w in kwargs.pop("opponent_wrappers", [])]
        unknown = [n for n in names if n not in ("OpponentPoolWrapper", "RandomOpponentWrapper", "RecordWinner")]
        if unknown:
            raise NotImplementedError("opponent_wrappers %s are not built for the device tic-tac-toe env" % unknown)
        if "OpponentPoolWrapper" in names:
            opponent = "pool"
        elif "RandomOpponentWrapper" in names:
            opponent = "random"
        if opponent == "pool":  # self-play against frozen snapshots of the learner
            return TicTacToeSelfPlayVecEnv(env_num, id, device=device, seed=seed, pool_size=kwargs.pop("pool_size", 4),
                                           opponent_sampling=kwargs.pop("opponent_sampling", "per_reset"),
                                           opponent_strategy=kwargs.pop("opponent_strategy", "RandomOpponent"))
        return TicTacToeVecEnv(env_num, id, device=device, seed=seed)
    raise NotImplementedError(
        "env id %r is not a device-resident env of the MI355X engine (built: 'CartPole-v1', 'simple_spread', 'tictactoe_v3', '%s-v0'); "
        "pass make_custom_envs=... returning a duck-typed VecEnv (gymnasium is not part of this engine)"
        % (id, _SYNTH_PREFIX))

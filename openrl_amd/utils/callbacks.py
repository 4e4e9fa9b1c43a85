"""SB3-style callback hooks the on-policy driver fires (``openrl/utils/callbacks/callbacks.py``):
``init_callback, on_training_start, on_rollout_start, update_locals, on_step, on_rollout_end,
on_training_end``.  ``on_step() is False`` aborts the rollout (onpolicy_driver.py:177-178)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union


class BaseCallback:
    #: a callback that needs per-env-step host control forces the stepwise rollout path
    needs_per_step = True

    def __init__(self, verbose: int = 0):
        self.agent = None
        self.n_calls = 0
        self.num_time_steps = 0
        self.verbose = verbose
        self.locals: Dict[str, Any] = {}
        self.globals: Dict[str, Any] = {}
        self.parent = None

    def init_callback(self, agent) -> None:
        self.agent = agent
        self._init_callback()

    def _init_callback(self) -> None:
        pass

    def on_training_start(self, locals_: Dict[str, Any], globals_: Dict[str, Any]) -> None:
        self.locals, self.globals = locals_, globals_
        self.num_time_steps = self.agent.num_time_steps
        self._on_training_start()

    def _on_training_start(self) -> None:
        pass

    def on_rollout_start(self) -> None:
        self._on_rollout_start()

    def _on_rollout_start(self) -> None:
        pass

    def _on_step(self) -> bool:
        return True

    def on_step(self) -> bool:
        self.n_calls += 1
        self.num_time_steps = self.agent.num_time_steps
        return self._on_step()

    def on_training_end(self) -> None:
        self._on_training_end()

    def _on_training_end(self) -> None:
        pass

    def on_rollout_end(self) -> None:
        self._on_rollout_end()

    def _on_rollout_end(self) -> None:
        pass

    def update_locals(self, locals_: Dict[str, Any]) -> None:
        self.locals.update(locals_)
        self.update_child_locals(locals_)

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        pass


class NoopCallback(BaseCallback):
    """Default when the user passes ``callback=None``: lets the driver use the fused rollout."""
    needs_per_step = False


class SelfPlayCallback(BaseCallback):
    """Every ``push_every`` rollouts, store a snapshot of the learner's policy in the env's opponent pool
    (``TicTacToeSelfPlayVecEnv.push_opponent``) - the role of the reference's selfplay callbacks / OpponentPoolWrapper."""
    needs_per_step = False

    def __init__(self, push_every: int = 10, verbose: int = 0):
        super().__init__(verbose)
        self.push_every, self.rollouts = int(push_every), 0

    def _on_rollout_start(self) -> None:
        env = self.agent.net.env if getattr(self.agent, "net", None) is not None else None
        if env is not None and hasattr(env, "push_opponent") and self.rollouts % self.push_every == 0 and self.rollouts:
            env.push_opponent(self.agent.net.module.models["policy"].theta)
        self.rollouts += 1


class ConvertCallback(BaseCallback):
    def __init__(self, callback: Callable[[Dict[str, Any], Dict[str, Any]], bool], verbose: int = 0):
        super().__init__(verbose)
        self.callback = callback

    def _on_step(self) -> bool:
        if self.callback is not None:
            return self.callback(self.locals, self.globals)
        return True


class CallbackList(BaseCallback):
    def __init__(self, callbacks: List[BaseCallback]):
        super().__init__()
        self.callbacks = callbacks
        self.needs_per_step = any(c.needs_per_step for c in callbacks)

    def _init_callback(self) -> None:
        for c in self.callbacks:
            c.init_callback(self.agent)

    def _on_training_start(self) -> None:
        for c in self.callbacks:
            c.on_training_start(self.locals, self.globals)

    def _on_rollout_start(self) -> None:
        for c in self.callbacks:
            c.on_rollout_start()

    def _on_step(self) -> bool:
        ok = True
        for c in self.callbacks:
            ok = c.on_step() and ok
        return ok

    def _on_rollout_end(self) -> None:
        for c in self.callbacks:
            c.on_rollout_end()

    def _on_training_end(self) -> None:
        for c in self.callbacks:
            c.on_training_end()

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        for c in self.callbacks:
            c.update_locals(locals_)


class EveryNTimesteps(BaseCallback):
    """Trigger ``callback`` every ``n_steps`` timesteps (openrl/utils/callbacks/callbacks.py EveryNTimesteps)."""

    def __init__(self, n_steps: int, callback: BaseCallback):
        super().__init__()
        self.n_steps, self.last_time_trigger, self.callback = int(n_steps), 0, callback

    def _init_callback(self) -> None:
        self.callback.parent = self
        self.callback.init_callback(self.agent)

    def _on_step(self) -> bool:
        if (self.num_time_steps - self.last_time_trigger) >= self.n_steps:
            self.last_time_trigger = self.num_time_steps
            return self.callback.on_step()
        return True

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        self.callback.update_locals(locals_)


class CheckpointCallback(BaseCallback):
    """``agent.save(save_path/{name_prefix}_{num_time_steps}_steps)`` every ``save_freq`` calls
    (openrl/utils/callbacks/checkpoint_callback.py:28-100; replay-buffer checkpoints are not built)."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rl_model", save_replay_buffer: bool = False,
                 verbose: int = 0):
        super().__init__(verbose)
        self.save_freq, self.save_path, self.name_prefix = int(save_freq), str(save_path), name_prefix
        self.saved: List[str] = []

    def _init_callback(self) -> None:
        import os

        os.makedirs(self.save_path, exist_ok=True)

    def _on_step(self) -> bool:
        if self.n_calls % self.save_freq == 0:
            import os

            path = os.path.join(self.save_path, "%s_%d_steps" % (self.name_prefix, self.num_time_steps))
            self.agent.save(path)
            self.saved.append(path)
            if self.verbose >= 2:
                print("Saving model checkpoint to %s" % path)
        return True


class StopTrainingOnMaxEpisodes(BaseCallback):
    """Stop once ``max_episodes`` episodes per env have finished (stop_callback.py:60-100): counts ``dones`` of the
    driver's locals, so it forces the stepwise rollout."""

    def __init__(self, max_episodes: int, verbose: int = 0):
        super().__init__(verbose)
        self.max_episodes, self.n_episodes, self._total = int(max_episodes), 0, int(max_episodes)

    def _init_callback(self) -> None:
        env = getattr(getattr(self.agent, "net", None), "env", None) or getattr(self.agent, "_env", None)
        self._total = self.max_episodes * int(getattr(env, "parallel_env_num", 1))

    def _on_step(self) -> bool:
        assert "dones" in self.locals, "`dones` variable is not defined next to callback.on_step()"
        d = self.locals["dones"]
        self.n_episodes += int(d.sum().item() if hasattr(d, "sum") else sum(d))
        return self.n_episodes < self._total


class ProgressBarCallback(BaseCallback):
    """Placeholder for the reference's rich progress bar (processbar_callback.py): counts steps, draws nothing."""
    needs_per_step = False


CALLBACKS = {"CheckpointCallback": CheckpointCallback, "StopTrainingOnMaxEpisodes": StopTrainingOnMaxEpisodes,
             "ProgressBarCallback": ProgressBarCallback, "EveryNTimesteps": EveryNTimesteps,
             "SelfPlayCallback": SelfPlayCallback}


class CallbackFactory:
    """``cfg.callbacks`` entries ``{"id": ..., "args": {...}}`` -> callback objects (callbacks_factory.py:14-60).
    EvalCallback / StopTrainingOnRewardThreshold / StopTrainingOnNoModelImprovement / SelfplayAPI are not built."""

    @staticmethod
    def get_callback(spec: Dict[str, Any]) -> BaseCallback:
        if spec["id"] not in CALLBACKS:
            raise ValueError("Callback %s not found (built: %s)" % (spec["id"], ", ".join(sorted(CALLBACKS))))
        return CALLBACKS[spec["id"]](**spec.get("args", {}))

    @staticmethod
    def get_callbacks(specs) -> "CallbackList":
        if isinstance(specs, dict):
            specs = [specs]
        return CallbackList([CallbackFactory.get_callback(s) for s in specs])


MaybeCallback = Union[None, Callable, List[BaseCallback], BaseCallback]


def as_callback(callback: MaybeCallback) -> BaseCallback:
    """rl_agent.py:132-164: list -> CallbackList, function -> ConvertCallback, None -> no-op."""
    if callback is None:
        return NoopCallback()
    if isinstance(callback, list):
        return CallbackList(callback)
    if not isinstance(callback, BaseCallback):
        return ConvertCallback(callback)
    return callback

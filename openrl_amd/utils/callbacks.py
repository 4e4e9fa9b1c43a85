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

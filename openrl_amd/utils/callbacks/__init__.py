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
        self.training_env = None
        self.n_calls = 0
        self.num_time_steps = 0
        self.verbose = verbose
        self.locals: Dict[str, Any] = {}
        self.globals: Dict[str, Any] = {}
        self.parent = None

    def init_callback(self, agent) -> None:
        self.agent = agent
        get_env = getattr(agent, "get_env", None)
        self.training_env = get_env() if callable(get_env) else getattr(agent, "_env", None)
        self.logger = getattr(agent, "logger", None)
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

    def set_parent(self, parent: "BaseCallback") -> None:
        self.parent = parent


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
    def __init__(self, callbacks: List[BaseCallback], stop_logic: str = "OR"):
        super().__init__()
        assert isinstance(callbacks, list)
        if stop_logic not in ("OR", "AND"):
            raise ValueError("Unknown stop logic %s, possible values are 'OR' or 'AND'" % (stop_logic,))
        self.callbacks, self.stop_logic = callbacks, stop_logic
        self.needs_per_step = any(callback_needs_per_step(c) for c in callbacks)

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
        # callbacks.py:200-222: "OR" stops when ANY child asks to, "AND" only when ALL do; every child always runs
        stops = [not c.on_step() for c in self.callbacks]
        return not (any(stops) if self.stop_logic == "OR" else all(stops))

    def _on_rollout_end(self) -> None:
        for c in self.callbacks:
            c.on_rollout_end()

    def _on_training_end(self) -> None:
        for c in self.callbacks:
            c.on_training_end()

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        for c in self.callbacks:
            c.update_locals(locals_)

    def set_parent(self, parent: "BaseCallback") -> None:
        self.parent = parent
        for c in self.callbacks:
            c.set_parent(parent)

    def __repr__(self):
        return str([type(c).__name__ for c in self.callbacks])


def _as_child(callbacks, stop_logic: str = "OR"):
    """A child given as ``{"id": ...}`` / a list of such specs goes through the factory (callbacks.py:296-299,
    eval_callback.py:102-114); a callback object is used as it is."""
    if isinstance(callbacks, (dict, list)):
        return CallbackFactory.get_callbacks(callbacks, stop_logic=stop_logic)
    return callbacks


class EventCallback(BaseCallback):
    """Base of the callbacks that trigger a child ``callback`` on an event (callbacks.py:133-173)."""

    def __init__(self, callback: Optional[BaseCallback] = None, verbose: int = 0):
        super().__init__(verbose=verbose)
        self.callback = callback
        if callback is not None:
            self.callback.set_parent(self)

    def init_callback(self, agent) -> None:
        super().init_callback(agent)
        if self.callback is not None:
            self.callback.init_callback(self.agent)

    def _on_training_start(self) -> None:
        if self.callback is not None:
            self.callback.on_training_start(self.locals, self.globals)

    def _on_event(self) -> bool:
        return self.callback.on_step() if self.callback is not None else True

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        if self.callback is not None:
            self.callback.update_locals(locals_)


class EveryNTimesteps(EventCallback):
    """Trigger ``callbacks`` every ``n_steps`` timesteps (callbacks.py:282-308)."""

    def __init__(self, n_steps: int, callbacks=None, stop_logic: str = "OR", callback: Optional[BaseCallback] = None):
        super().__init__(_as_child(callbacks if callbacks is not None else callback, stop_logic))
        self.n_steps, self.last_time_trigger = int(n_steps), 0

    def _on_step(self) -> bool:
        if (self.num_time_steps - self.last_time_trigger) >= self.n_steps:
            self.last_time_trigger = self.num_time_steps
            return self._on_event()
        return True


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


class StopTrainingOnRewardThreshold(BaseCallback):
    """Child of ``EvalCallback`` (``callbacks_on_new_best``): stop once the best mean evaluation reward reaches the
    threshold (stop_callback.py:25-57)."""

    def __init__(self, reward_threshold: float, verbose: int = 0):
        super().__init__(verbose)
        self.reward_threshold = float(reward_threshold)

    def _on_step(self) -> bool:
        assert self.parent is not None, "StopTrainingOnRewardThreshold must be used with an EvalCallback"
        return bool(self.parent.best_mean_reward < self.reward_threshold)


class StopTrainingOnNoModelImprovement(BaseCallback):
    """Child of ``EvalCallback`` (``callbacks_after_eval``): stop when ``max_no_improvement_evals`` consecutive
    evaluations brought no new best mean reward, counting only after ``min_evals`` evaluations
    (stop_callback.py:107-154)."""

    def __init__(self, max_no_improvement_evals: int, min_evals: int = 0, verbose: int = 1):
        super().__init__(verbose)
        self.max_no_improvement_evals, self.min_evals = int(max_no_improvement_evals), int(min_evals)
        self.last_best_mean_reward = -float("inf")
        self.no_improvement_evals = 0

    def _on_step(self) -> bool:
        assert self.parent is not None, "StopTrainingOnNoModelImprovement must be used with an EvalCallback"
        continue_training = True
        if self.n_calls > self.min_evals:
            if self.parent.best_mean_reward > self.last_best_mean_reward:
                self.no_improvement_evals = 0
            else:
                self.no_improvement_evals += 1
                if self.no_improvement_evals > self.max_no_improvement_evals:
                    continue_training = False
        self.last_best_mean_reward = self.parent.best_mean_reward
        if self.verbose >= 1 and not continue_training:
            print("Stopping training because there was no new best model in the last %d evaluations"
                  % self.no_improvement_evals)
        return continue_training


class EvalCallback(EventCallback):
    """Every ``eval_freq`` calls, play ``n_eval_episodes`` episodes per env of a separate DEVICE-RESIDENT evaluation env
    with the current policy (greedy by default), track the best mean episode return, save the best model and run the
    ``callbacks_on_new_best`` / ``callbacks_after_eval`` children (openrl/utils/callbacks/eval_callback.py:38-283; same
    constructor).  ``eval_env``: a device VecEnv of this package, an env id, or the ``{"id": ..., "env_num": ...}``
    spec ``make`` takes.  Episode returns come from the env's in-kernel episode statistics, so the evaluation never
    leaves the device; what ``log_path``/evaluations.npz holds per evaluation is therefore one mean return and one mean
    length PER ENV LANE, not one entry per episode.  ``render`` / ``asynchronous`` / ``warn`` are accepted for signature
    compatibility and have nothing to act on here (no renderer, no worker processes, no Monitor wrapper)."""

    def __init__(self, eval_env, callbacks_on_new_best=None, callbacks_after_eval=None, n_eval_episodes: int = 5,
                 eval_freq: int = 10000, log_path=None, best_model_save_path=None, deterministic: bool = True,
                 render: bool = False, asynchronous: bool = True, verbose: int = 1, warn: bool = True,
                 stop_logic: str = "OR", close_env_at_end: bool = True, max_eval_steps: int = 100000):
        import os

        super().__init__(_as_child(callbacks_after_eval, stop_logic), verbose=verbose)
        self.stop_logic = stop_logic
        self.callbacks_on_new_best = _as_child(callbacks_on_new_best, stop_logic)
        if self.callbacks_on_new_best is not None:
            self.callbacks_on_new_best.set_parent(self)
        self.n_eval_episodes, self.eval_freq, self.deterministic = int(n_eval_episodes), int(eval_freq), bool(deterministic)
        self.render, self.warn, self.close_env_at_end = render, warn, close_env_at_end
        self.best_mean_reward, self.last_mean_reward = -float("inf"), -float("inf")
        self.eval_env_spec, self.eval_env = eval_env, None
        self.best_model_save_path = None if best_model_save_path is None else str(best_model_save_path)
        self.log_path = None if log_path is None else os.path.join(str(log_path), "evaluations")
        self.max_eval_steps = int(max_eval_steps)
        self.evaluations_results, self.evaluations_time_steps, self.evaluations_length = [], [], []
        self.evaluations = []  # (num_time_steps, mean episode return, episodes)

    # the names used before the constructor followed the reference's
    on_new_best = property(lambda self: self.callbacks_on_new_best)
    after_eval = property(lambda self: self.callback)

    def _init_callback(self) -> None:
        import os

        if self.eval_env is None:
            if isinstance(self.eval_env_spec, (dict, str)):
                from ...envs.common import make

                spec = dict(self.eval_env_spec) if isinstance(self.eval_env_spec, dict) else {"id": self.eval_env_spec}
                self.eval_env = make(spec.pop("id"), **spec)
            else:
                self.eval_env = self.eval_env_spec
        if not getattr(self.eval_env, "is_device_env", False):
            raise NotImplementedError("EvalCallback evaluates on the device-resident envs of this package")
        if self.best_model_save_path is not None:
            os.makedirs(self.best_model_save_path, exist_ok=True)
        if self.log_path is not None:
            os.makedirs(os.path.dirname(self.log_path), exist_ok=True)
        if self.callbacks_on_new_best is not None:
            self.callbacks_on_new_best.init_callback(self.agent)

    def evaluate_lanes(self):
        """-> (mean finished-episode return per lane [n], mean finished-episode length per lane [n], episodes)."""
        import torch

        env, module = self.eval_env, self.agent.net.module
        n, a = env.parallel_env_num, env.agent_num
        obs = env.reset_device(seed=getattr(env, "seed", 0))
        env.ep_stats.zero_()
        target = self.n_eval_episodes * n
        h = None
        if getattr(module, "recurrent", False):
            # one row of recurrent state per lane: [recurrent_N x (H | 2 H for an LSTM)] on the general towers, H on the
            # default one
            pn = getattr(module, "policy_net", None)
            width = pn.state_w * pn.recurrent_N if pn is not None and hasattr(pn, "state_w") else module.cfg.hidden_size
            h = torch.zeros(n * a, width, device=env.device)
        masks = torch.ones(n * a, 1, device=env.device)
        run_len = torch.zeros(n, device=env.device)
        fin_len = torch.zeros(n, device=env.device)
        for step in range(self.max_eval_steps):
            p_obs = obs["policy"] if isinstance(obs, dict) else obs
            am = getattr(env, "action_mask_device", None)
            act, h2 = module.act(p_obs.reshape(n * a, -1), h, masks, action_masks=None if am is None else am.reshape(n * a, -1),
                                 deterministic=self.deterministic)
            obs, _, done = env.step_device(act.view(n, a, -1))
            d0 = done.reshape(n, -1)[:, 0].float()
            run_len += 1.0
            fin_len += run_len * d0
            run_len *= 1.0 - d0
            if h is not None:
                masks = 1.0 - done.reshape(n * a, 1).float()
                h = h2.reshape(n * a, -1)
            if step % 8 == 7 and env.episode_statistics()["episodes_finished"] >= target:
                break
        st = env.ep_stats.reshape(n, -1, 4)[:, 0].double().cpu()
        cnt = st[:, 3].clamp_min(1.0)
        return (st[:, 2] / cnt).numpy(), (fin_len.double().cpu() / cnt).numpy(), int(st[:, 3].sum().item())

    def evaluate(self):
        st = self.evaluate_lanes()
        return self.eval_env.episode_statistics()["episode_return_mean"], st[2]

    def _on_step(self) -> bool:
        if self.eval_freq <= 0 or self.n_calls % self.eval_freq != 0:
            return True
        import os

        import numpy as np

        lane_ret, lane_len, episodes = self.evaluate_lanes()
        mean_reward = float(self.eval_env.episode_statistics()["episode_return_mean"])
        self.last_mean_reward = mean_reward
        self.evaluations.append((self.num_time_steps, mean_reward, episodes))
        if self.log_path is not None:
            self.evaluations_time_steps.append(self.num_time_steps)
            self.evaluations_results.append(lane_ret)
            self.evaluations_length.append(lane_len)
            np.savez(self.log_path, timesteps=self.evaluations_time_steps, results=self.evaluations_results,
                     ep_lengths=self.evaluations_length)
        if self.verbose >= 1:
            print("Eval num_timesteps=%d, episode_reward=%.2f +/- %.2f over %d episodes"
                  % (self.num_time_steps, mean_reward, float(np.std(lane_ret)), episodes))
            print("Episode length: %.2f +/- %.2f" % (float(np.mean(lane_len)), float(np.std(lane_len))))
        keep = True
        if mean_reward > self.best_mean_reward:
            if self.verbose >= 1:
                print("New best mean reward!")
            if self.best_model_save_path is not None:
                self.agent.save(os.path.join(self.best_model_save_path, "best_model"))
                with open(os.path.join(self.best_model_save_path, "best_model_info.txt"), "w") as f:
                    f.write("best model at step: %d\n" % self.num_time_steps)
                    f.write("best model reward: %s\n" % mean_reward)
            self.best_mean_reward = mean_reward
            if self.callbacks_on_new_best is not None:
                keep = self.callbacks_on_new_best.on_step()
        if self.callback is not None:
            keep = keep and self._on_event()
        return keep

    def update_child_locals(self, locals_: Dict[str, Any]) -> None:
        for child in (self.callbacks_on_new_best, self.callback):
            if child is not None:
                child.update_locals(locals_)

    def _on_training_end(self) -> None:
        if self.close_env_at_end and self.eval_env is not None:
            self.eval_env.close()


CALLBACKS = {"CheckpointCallback": CheckpointCallback, "EvalCallback": EvalCallback,
             "StopTrainingOnRewardThreshold": StopTrainingOnRewardThreshold, "StopTrainingOnMaxEpisodes": StopTrainingOnMaxEpisodes,
             "StopTrainingOnNoModelImprovement": StopTrainingOnNoModelImprovement,
             "ProgressBarCallback": ProgressBarCallback, "EveryNTimesteps": EveryNTimesteps,
             "SelfPlayCallback": SelfPlayCallback}
callbacks_dict = CALLBACKS  # the reference's name for the registry (callbacks_factory.py:14)


class CallbackFactory:
    """``cfg.callbacks`` entries ``{"id": ..., "args": {...}}`` -> callback objects (callbacks_factory.py:14-60).
    The SelfplayAPI callbacks (they talk to the reference's self-play HTTP service) are not built."""

    @staticmethod
    def get_callback(spec: Dict[str, Any]) -> BaseCallback:
        if spec["id"] not in CALLBACKS:
            raise ValueError("Callback %s not found (built: %s)" % (spec["id"], ", ".join(sorted(CALLBACKS))))
        return CALLBACKS[spec["id"]](**spec.get("args", {}))

    @staticmethod
    def get_callbacks(specs, stop_logic: str = "OR") -> "CallbackList":
        if isinstance(specs, dict):
            specs = [specs]
        return CallbackList([CallbackFactory.get_callback(s) for s in specs], stop_logic=stop_logic)

    @staticmethod
    def register(id: str, callback_class) -> None:
        CALLBACKS[id] = callback_class


MaybeCallback = Union[None, Callable, List[BaseCallback], BaseCallback]


_HOOKS = ("init_callback", "on_rollout_start", "update_locals", "on_step", "on_rollout_end")


def is_callback_object(cb) -> bool:
    """Duck type of a callback OBJECT: this package's ``BaseCallback`` or a foreign one with the same hooks - e.g. the
    reference's own ``openrl.utils.callbacks.callbacks.{CallbackList,ConvertCallback}`` that ``RLAgent._init_callback``
    (rl_agent.py:137-164) always hands a driver injected through ``driver_class``, even for ``callback=None``."""
    return isinstance(cb, BaseCallback) or all(callable(getattr(cb, h, None)) for h in _HOOKS)


def callback_needs_per_step(cb) -> bool:
    """Does this callback object need host control at every env step (forces the stepwise rollout)?  Foreign objects
    carry no ``needs_per_step`` attribute: the reference's wrappers around "no callback" (``ConvertCallback(None)``,
    an empty or all-no-op ``CallbackList``) do not, anything else is assumed to."""
    flag = getattr(cb, "needs_per_step", None)
    if flag is not None and not isinstance(cb, CallbackList):
        return bool(flag)
    children = getattr(cb, "callbacks", None)
    if isinstance(children, (list, tuple)):  # CallbackList (ours or foreign)
        return any(callback_needs_per_step(c) for c in children)
    if type(cb).__name__ == "ConvertCallback" and getattr(cb, "callback", 0) is None:
        return False
    return True if flag is None else bool(flag)


def as_callback(callback: MaybeCallback):
    """rl_agent.py:132-164: list -> CallbackList, function -> ConvertCallback, None -> no-op; a callback object
    (ours or a foreign duck type, see ``is_callback_object``) is used as it is."""
    if callback is None:
        return NoopCallback()
    if isinstance(callback, list):
        return CallbackList(callback)
    if is_callback_object(callback):
        return callback
    return ConvertCallback(callback)

"""``OnPolicyDriver`` - the rollout / update loop (``openrl/drivers/rl_driver.py:28-180`` +
``openrl/drivers/onpolicy_driver.py:32-279``) for the MI355X engine.

Two rollout paths behind the same ``actor_rollout``:

* **fused** (device-resident envs, no per-step callback): ONE launch of ``orl_rollout_fused`` runs all
  ``episode_length`` steps of {policy+value forward, sample, env.step, buffer insert, bootstrap value};
  nothing crosses PCIe.  The reference crosses the device boundary twice per step and spends most of
  its rollout time in ``np.split`` / ``np.concatenate`` (K16).
* **stepwise** (any duck-typed VecEnv, or callbacks that need ``on_step``): per step one ``orl_act_step``
  writing values / actions / log-probs straight into the buffer slot, ``env.step``, one
  ``orl_buffer_insert`` that also builds masks / active_masks / bad_masks on the device
  (onpolicy_driver.py:91-138).

Callback protocol, ``agent.num_time_steps`` accounting and the ``({}, False)`` abort are those of
onpolicy_driver.py:155-196.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .. import ops, ops_rnn
from ..utils.callbacks import as_callback, callback_needs_per_step
from ..utils.logger import Logger


def prepare_action_masks(infos, agent_num: int):
    """envs/vec_env/utils/util.py:54-88 (as_batch=False): ``info["action_masks"]`` per env -> [N, A, K] or None."""
    if infos is None:
        return None
    if isinstance(infos, dict):
        return np.asarray(infos["action_masks"]) if "action_masks" in infos else None
    if len(infos) == 0 or not isinstance(infos[0], dict) or "action_masks" not in infos[0]:
        return None
    return np.stack([np.asarray(i["action_masks"]).reshape(agent_num, -1) for i in infos])


class HostStaging:
    """Pinned-memory staging for HOST envs (gymnasium / PettingZoo / Isaac wrappers behind the VecEnv duck type):
    every array that crosses PCIe per step - observations, rewards, dones, masks in; actions out - goes through a
    page-locked buffer allocated once per (name, shape, dtype), so the copies are DMA transfers queued on the launch
    stream instead of pageable staged copies that block the host.  Safe to reuse the
    buffers step after step: a step ends with the synchronising device -> host copy of the actions, which orders every
    earlier host -> device copy of the stream before the host touches the pinned arrays again."""

    def __init__(self, device) -> None:
        self.device = device
        self._in, self._out = {}, {}

    def to_device(self, name: str, arr, dtype: torch.dtype) -> torch.Tensor:
        a = np.asarray(arr)
        key = (name, a.shape, dtype)
        pair = self._in.get(key)
        if pair is None:
            pin = torch.empty(a.shape, dtype=dtype).pin_memory()
            pair = self._in[key] = (pin, pin.numpy(), torch.empty(a.shape, dtype=dtype, device=self.device))
        pin, view, dev = pair
        np.copyto(view, a, casting="unsafe")  # one host pass: dtype conversion straight into the pinned pages
        dev.copy_(pin, non_blocking=True)
        return dev

    def to_device_many(self, items):
        """``items``: [(name, array, torch dtype)] of one env step -> {name: device tensor}, through ONE pinned byte
        buffer and ONE host -> device copy (each ``copy_`` costs the host ~6 us; a step brings 3-6 arrays)."""
        arrs = [(n, np.asarray(a), dt) for n, a, dt in items]
        key = tuple((n, a.shape, dt) for n, a, dt in arrs)
        pack = self._in.get(key)
        if pack is None:
            offs, total = [], 0
            for _, a, dt in arrs:
                offs.append(total)
                nbytes = int(np.prod(a.shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
                total += (nbytes + 15) // 16 * 16
            pin = torch.empty(max(total, 16), dtype=torch.uint8).pin_memory()
            dev = torch.empty(max(total, 16), dtype=torch.uint8, device=self.device)
            host_views, dev_views = [], {}
            for (n, a, dt), off in zip(arrs, offs):
                nbytes = int(np.prod(a.shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
                host_views.append(pin[off:off + nbytes].view(dt).view(a.shape).numpy())
                dev_views[n] = dev[off:off + nbytes].view(dt).view(a.shape)
            pack = self._in[key] = (pin, dev, host_views, dev_views)
        pin, dev, host_views, dev_views = pack
        for (_, a, _), view in zip(arrs, host_views):
            np.copyto(view, a, casting="unsafe")
        dev.copy_(pin, non_blocking=True)
        return dev_views

    def to_host(self, name: str, t: torch.Tensor) -> np.ndarray:
        key = (name, tuple(t.shape), t.dtype)
        pin = self._out.get(key)
        if pin is None:
            pin = self._out[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        pin.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin.numpy()


class OnPolicyDriver:
    #: the buffer this driver works on (device-resident); see INTEGRATION.md section 2
    from ..buffers import NormalReplayBuffer as buffer_class

    def __init__(self, config: Dict[str, Any], trainer, buffer, agent, rank: int = 0, world_size: int = 1, client=None,
                 logger: Optional[Logger] = None, callback=None) -> None:
        self.trainer = trainer
        self.buffer = buffer
        self.world_size = world_size
        self.rank = rank
        self.logger = logger if logger is not None else Logger()
        cfg = config["cfg"]
        self.cfg = cfg
        self.envs = config["envs"]
        self.device = nat.require_gpu(config["device"])
        self.callback = as_callback(callback)
        self.agent = agent
        if getattr(self.callback, "agent", None) is None:
            self.callback.init_callback(agent)
        per_step_cb = callback_needs_per_step(self.callback)
        self.num_agents = config["num_agents"]
        self.num_env_steps = cfg.num_env_steps
        self.episode_length = cfg.episode_length
        self.n_rollout_threads = cfg.n_rollout_threads
        self.learner_n_rollout_threads = cfg.learner_n_rollout_threads
        self.use_linear_lr_decay = cfg.use_linear_lr_decay
        self.hidden_size = cfg.hidden_size
        self.recurrent_N = cfg.recurrent_N
        self.log_interval = cfg.log_interval
        self.episode = 0
        self.episodes = 0
        self.total_num_steps = 0
        mode = getattr(cfg, "amd_rollout_mode", "auto")
        dev_env = bool(getattr(self.envs, "is_device_env", False))
        if mode == "fused" and not dev_env:
            raise ValueError("amd_rollout_mode=fused needs a device-resident env")
        recurrent = bool(getattr(self.trainer.algo_module, "recurrent", False))
        generic = bool(getattr(self.trainer.algo_module, "generic", False))  # general towers roll out stepwise
        # general towers: the fused rollout (orl_gen_rollout_fused) on the device-resident single-agent envs, feed-forward
        self.fused_generic = (dev_env and generic and not recurrent
                              and getattr(self.envs, "env_kind", None) in (ops.ENV_SYNTH, ops.ENV_CARTPOLE)
                              and self.trainer.algo_module.fused_rollout_ready(self.buffer.data))
        if mode == "fused" and generic and not self.fused_generic:
            raise ValueError("amd_rollout_mode=fused with general towers: feed-forward towers of widths <= 256 on the "
                             "device synthetic / CartPole envs")
        # recurrent policies: fused on envs orl_rnn_rollout_fused steps in-kernel (the device MPE simple_spread)
        # ... and on the single-agent device envs (synthetic, CartPole: one shared observation array of width <= 64)
        single_rnn = (getattr(self.envs, "env_kind", None) in (ops.ENV_SYNTH, ops.ENV_CARTPOLE)
                      and self.buffer.data.num_agents == 1 and self.buffer.data.critic_obs is self.buffer.data.policy_obs
                      and self.buffer.data.Dp <= 64)
        self.fused_rnn = (dev_env and recurrent and not generic
                          and (bool(getattr(self.envs, "supports_fused_rnn_rollout", False)) or single_rnn))
        self._fused_rnn_single = self.fused_rnn and single_rnn
        if mode == "fused" and recurrent and not self.fused_rnn:
            raise ValueError("amd_rollout_mode=fused with a recurrent policy is built for the device MPE env only")
        can_fuse = self.fused_rnn or self.fused_generic or (
            dev_env and bool(getattr(self.envs, "supports_fused_rollout", True)) and not recurrent and not generic)
        if mode == "fused" and not can_fuse:
            raise ValueError("amd_rollout_mode=fused is not built for env %r" % getattr(self.envs, "env_name", "?"))
        self.fused = can_fuse and mode in ("auto", "fused") and not per_step_cb
        # stepwise rollouts of device envs whose step takes no per-call host scalar are captured once into a hipGraph
        # (T x {act, env.step, insert} ~ 125 launches) and replayed; the Philox step counter lives on the device
        self._graph_ok = (dev_env and not self.fused and bool(getattr(cfg, "amd_use_graph", True))
                          and bool(getattr(self.envs, "supports_graph_rollout", False))
                          and not per_step_cb)
        self._graph = None
        self._rng_ctr = None
        self._chase_flags = None  # orl_rnn_rollout_fused step counters (recurrent fused rollout)
        self._chase_watch = None
        self._staging = None if dev_env else HostStaging(self.device)
        d = self.buffer.data
        self._next_value = torch.zeros(d.n_rollout_threads, d.num_agents, 1, dtype=torch.float32, device=self.device)
        self._have_next_value = False

    # ---------------------------------------------------------------------------------------- rl_driver.py
    def reset_and_buffer_init(self):
        if getattr(self.envs, "is_device_env", False):
            obs = self.envs.reset_device(seed=getattr(self.envs, "seed", None))
            info = None
            dev_masks = getattr(self.envs, "action_mask_device", None)  # legal-move masks that never leave the device
            if dev_masks is not None:
                self.buffer.init_buffer(obs, action_masks=dev_masks)
                return
        else:
            returns = self.envs.reset()
            if isinstance(returns, tuple):
                assert len(returns) == 2, "length of env reset returns must be 2, but get {}".format(len(returns))
                obs, info = returns
            else:
                obs, info = returns, None
        self.buffer.init_buffer(obs, action_masks=prepare_action_masks(info, self.num_agents))

    def run(self) -> None:
        episodes = int(self.num_env_steps) // self.episode_length // self.learner_n_rollout_threads
        self.episodes = episodes
        self.reset_and_buffer_init()
        for episode in range(episodes):
            self.episode = episode
            if not self._inner_loop():
                break
        self.check_device_errors()

    def check_device_errors(self) -> None:
        """Raise if a kernel set an error word during the run (waits for the last posted copies)."""
        for w in (self._chase_watch, getattr(self.trainer, "_comm_watch", None)):
            if w is not None:
                w.poll(wait=True)

    def learner_update(self):
        if self.use_linear_lr_decay:
            self.trainer.algo_module.lr_decay(self.episode, self.episodes)
        self.compute_returns()
        self.trainer.prep_training()
        return self.trainer.train(self.buffer.data)

    # ---------------------------------------------------------------------------------------- onpolicy_driver.py
    def _inner_loop(self) -> bool:
        rollout_infos, continue_training = self.actor_rollout()
        if not continue_training:
            return False
        train_infos = self.learner_update()
        self.buffer.after_update()
        self.total_num_steps = (self.episode + 1) * self.episode_length * self.n_rollout_threads
        if self.episode % self.log_interval == 0:
            self.logger.log_info(rollout_infos, step=self.total_num_steps)
            self.logger.log_info(train_infos, step=self.total_num_steps)
        return True

    def actor_rollout(self) -> Tuple[Dict[str, Any], bool]:
        self.callback.on_rollout_start()
        if hasattr(self.envs, "on_rollout_start"):  # e.g. the self-play env's per-rollout opponent draw
            self.envs.on_rollout_start()
        self.trainer.prep_rollout()
        if self.fused:
            self._fused_rollout()
        elif self._graph_ok and self.episode >= 1:  # the first rollout runs eagerly (allocations, module loads)
            self._graph_rollout()
        else:
            for step in range(self.episode_length):
                if not self._rollout_step(step):
                    return {}, False
        batch_rew_infos = self.envs.batch_rewards(self.buffer)
        self.callback.on_rollout_end()
        if getattr(self.envs, "use_monitor", False):
            if self.episode % self.log_interval == 0:
                statistics_info = self.envs.statistics(self.buffer)
            else:  # keep the FPS step count without a device->host sync
                statistics_info = {}
                if hasattr(self.envs, "count_steps"):
                    self.envs.count_steps(self.buffer)
                else:
                    statistics_info = self.envs.statistics(self.buffer)
            statistics_info.update(batch_rew_infos)
            return statistics_info, True
        return batch_rew_infos, True

    def _fused_rollout(self) -> None:
        d = self.buffer.data
        mod = self.trainer.algo_module
        if self.fused_generic:  # general towers: policy + sampling + env + insert in one launch, the critic in a second
            env = self.envs
            mod.rollout_fused(d, env, self._next_value)
            env.global_step += self.episode_length
            self._have_next_value = not mod.share_model
            d.step = 0
            d._adv_fresh = False
            self.agent.num_time_steps += env.parallel_env_num * self.episode_length
            return
        p, c = mod.models["policy"], mod.models["critic"]
        env = self.envs
        f = nat.fptr
        if self.fused_rnn:  # recurrent policy: policy + env launch, then the critic sweep (orl_rnn_rollout_fused)
            a = nat.RnnRolloutArgs()
            a.buf = d.buffer_ptrs()
            a.value_preds, a.actions, a.action_log_probs = f(d.value_preds), f(d.actions), f(d.action_log_probs)
            a.rnn_states, a.rnn_states_critic = f(d.rnn_states), f(d.rnn_states_critic)
            a.env_state, a.ep_stats = f(env.env_state), f(env.ep_stats)
            if not self._fused_rnn_single:
                a.obs_policy_out, a.obs_critic_out = f(env.obs["policy"]), f(env.obs["critic"])
            a.next_value = f(self._next_value)
            a.env_kind, a.world_length, a.deterministic = env.env_kind, env.episode_limit, 0
            a.env_seed, a.act_seed = env.seed & (2 ** 64 - 1), mod.act_seed & (2 ** 64 - 1)
            a.rng_step0 = int(mod.rng_step)
            a.env_step0 = int(env.global_step) & (2 ** 64 - 1)
            # critic in the same launch, one step behind (the MPE kernel; single-agent envs: policy launch + critic sweep)
            if bool(getattr(self.cfg, "amd_rnn_rollout_chase", True)) and not self._fused_rnn_single:
                if self._chase_flags is None:
                    self._chase_flags = torch.zeros((d.n_rollout_threads + 15) // 16 + 1, dtype=torch.int32,
                                                    device=self.device)
                a.sync_flags = nat.ptr(self._chase_flags)
            ops_rnn.rnn_rollout_fused(p.net, p.theta, c.net, c.theta, a, self.device)
            if self._chase_flags is not None and not self._fused_rnn_single:  # the kernel's error word: looked at (without a sync) next rollout / at the end
                if self._chase_watch is None:
                    self._chase_watch = nat.DeviceErrorWatch(
                        "orl_rnn_rollout_fused: a critic workgroup's bounded wait for its policy workgroup timed out - "
                        "the rollout's value predictions are incomplete")
                self._chase_watch.post(self._chase_flags[-1:])
            env.global_step += self.episode_length
            mod.rng_step += self.episode_length
            self._have_next_value = True
            d.step = 0
            d._adv_fresh = False
            self.agent.num_time_steps += env.parallel_env_num * self.episode_length
            return
        args = nat.RolloutArgs(d.buffer_ptrs(), f(d.value_preds), f(d.actions), f(d.action_log_probs), f(env.env_state),
                               f(env.ep_stats), env.env_kind, env.episode_limit, env.seed & (2 ** 64 - 1),
                               mod.act_seed & (2 ** 64 - 1), env.global_step)
        if hasattr(env, "fill_rollout_args"):  # env-specific extras (the self-play opponent pool)
            env.fill_rollout_args(args)
        if getattr(self.cfg, "amd_rollout_kernel", "chain") == "lockstep":  # the round-5 kernel (comparison switch)
            args.opp_reserved = 1
        ops.rollout_fused(p.net, p.theta, c.net, c.theta, args, self._next_value)
        if hasattr(env, "after_fused_rollout"):
            env.after_fused_rollout(self.episode_length)
        env.global_step += self.episode_length
        mod.rng_step += self.episode_length
        self._have_next_value = True
        d.step = 0
        d._adv_fresh = False
        self.agent.num_time_steps += env.parallel_env_num * self.episode_length

    def _graph_rollout(self) -> None:
        mod = self.trainer.algo_module
        T = self.episode_length
        if self._graph is None:
            self._rng_ctr = torch.full((1,), int(mod.rng_step), dtype=torch.int64, device=self.device)
            saved_steps, saved_rng = self.agent.num_time_steps, mod.rng_step
            mod.rng_step = 0  # the captured launches carry rng_step = 0 .. T-1; the device counter is the base
            mod.rng_step_dev = self._rng_ctr  # explicit argument of every act launch captured below
            self.envs.rng_step_dev = self._rng_ctr  # ... including the opponents' launches of a self-play env
            self.envs.rng_step_host = int(saved_rng)  # the counter's value now: envs with their own step count subtract it
            env_step0 = getattr(self.envs, "global_step", None)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    for step in range(T):
                        self._rollout_step(step)
                    self._rng_ctr.add_(T)
            finally:
                mod.rng_step_dev = None
                self.envs.rng_step_dev = None
                self.agent.num_time_steps, mod.rng_step = saved_steps, saved_rng
                if env_step0 is not None:  # capturing ran the Python side of T steps without executing them
                    self.envs.global_step = env_step0
            self._graph = graph
        self._graph.replay()
        mod.rng_step += T
        if getattr(self.envs, "global_step", None) is not None:
            self.envs.global_step += T
        self.agent.num_time_steps += self.envs.parallel_env_num * T
        d = self.buffer.data
        d.step = 0
        d._adv_fresh = False

    def _rollout_step(self, step: int) -> bool:
        d = self.buffer.data
        values, actions, action_log_probs, rnn_states, rnn_states_critic = self.act(step)
        extra_data = {"actions": actions, "values": values, "action_log_probs": action_log_probs, "step": step,
                      "buffer": self.buffer}
        if getattr(self.envs, "is_device_env", False):
            obs, rewards, dones = self.envs.step_device(actions)
            infos = None
        else:
            # host envs get the reference's action dtype: integer indices for Discrete / MultiDiscrete spaces (the
            # reference's ACTLayer samples int64, act.py:59-83), float32 for Box; the buffer keeps them as floats
            # (the float -> int64 conversion runs on the host copy: one device launch less per step)
            # a FRESH array every step, like the reference: envs / wrappers / callbacks may keep it (recording, replay),
            # and the pinned staging buffer is overwritten by the next step
            host_actions = self._staging.to_host("actions", actions)
            host_actions = host_actions.astype(np.int64) if self.buffer.data.act_is_index else host_actions.copy()
            obs, rewards, dones, infos = self.envs.step(host_actions, extra_data)
        self.agent.num_time_steps += self.envs.parallel_env_num
        self.callback.update_locals(locals())
        if self.callback.on_step() is False:
            return False
        self.add2buffer({"obs": obs, "rewards": rewards, "dones": dones, "infos": infos, "step": step,
                         "action_masks": getattr(self.envs, "action_mask_device", None)})
        return True

    @torch.no_grad()
    def act(self, step: int):
        """get_actions on slot ``step`` (onpolicy_driver.py:235-279); results land directly in the buffer
        slot ([N, A, .] views) - no np.split / np.concatenate round trips."""
        d = self.buffer.data
        mod = self.trainer.algo_module
        out = (d.value_preds[step].view(-1, 1), d.actions[step].view(-1, d.act_shape),
               d.action_log_probs[step].view(-1, d.act_shape))
        if getattr(mod, "recurrent", False):
            # new hidden states go straight into slot step+1 (ReplayData.insert, replay_data.py:262-263);
            # add2buffer zeroes them where the env finished (onpolicy_driver.py:91-108)
            H = d.hidden_size
            mod._forward_rnn(d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                             d.rnn_states[step].view(-1, H), d.rnn_states_critic[step].view(-1, H),
                             d.masks[step].view(-1), d.get_batch_data("action_masks", step), False, out=out,
                             h_out=(d.rnn_states[step + 1].view(-1, H), d.rnn_states_critic[step + 1].view(-1, H)))
            return (d.value_preds[step], d.actions[step], d.action_log_probs[step], d.rnn_states[step + 1],
                    d.rnn_states_critic[step + 1])
        mod._forward(d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                     d.get_batch_data("action_masks", step), False, out=out)
        return d.value_preds[step], d.actions[step], d.action_log_probs[step], None, None

    def _as_dev(self, x, dtype=torch.float32, name: str = ""):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype).contiguous()
        if self._staging is not None and name:  # host env: through the pinned staging buffers
            return self._staging.to_device(name, x, dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(self.device, dtype).contiguous()

    def add2buffer(self, data):
        """Mask construction + insert on the device (onpolicy_driver.py:80-152 -> orl_buffer_insert)."""
        d = self.buffer.data
        step = data["step"] if "step" in data else d.step
        obs = data["obs"]
        p_obs, c_obs = (obs.get("policy", obs), obs.get("critic", obs)) if isinstance(obs, dict) else (obs, obs)
        infos = data["infos"]
        bad = None
        if infos is not None and len(infos) and isinstance(infos[0], dict) and any("bad_transition" in i for i in infos):
            bad = np.array([[bool(i.get("bad_transition", [False] * self.num_agents)[a]) for a in range(self.num_agents)]
                            for i in infos], dtype=np.uint8)
        amask = data.get("action_masks")  # device envs: already a device tensor [N, A, K]
        if amask is None:
            amask = prepare_action_masks(infos, self.num_agents)
        same_obs = c_obs is obs or d.critic_obs is d.policy_obs
        fields = [("policy_obs", p_obs, torch.float32), ("critic_obs", None if same_obs else c_obs, torch.float32),
                  ("rewards", data["rewards"], torch.float32), ("dones", data["dones"], torch.uint8),
                  ("bad_transition", bad, torch.uint8), ("action_masks", amask, torch.float32)]
        dev = {}
        host = [(n, x, dt) for n, x, dt in fields if x is not None and not isinstance(x, torch.Tensor)]
        if self._staging is not None and host:  # host env: every numpy array of the step in ONE pinned copy
            dev = self._staging.to_device_many(host)
        for n, x, dt in fields:
            if x is not None and n not in dev:
                dev[n] = self._as_dev(x, dt)
        rec = d.rnn_states.stride(0) != 0  # recurrent: rnn_states[dones_env] = 0, folded into the insert launch
        ops.buffer_insert(d.buffer_ptrs(), step, dev["policy_obs"], dev.get("critic_obs", dev["policy_obs"]),
                          dev["rewards"], dev["dones"], dev.get("bad_transition"), dev.get("action_masks"),
                          d.rnn_states[step + 1] if rec else None, d.rnn_states_critic[step + 1] if rec else None)
        d.step = (step + 1) % d.episode_length
        d._adv_fresh = False

    @torch.no_grad()
    def compute_returns(self):
        self.trainer.prep_rollout()
        d = self.buffer.data
        mod = self.trainer.algo_module
        if self._have_next_value:
            next_values = self._next_value
            self._have_next_value = False
        else:
            rec = getattr(mod, "recurrent", False)
            next_values = mod.get_values(d.get_batch_data("critic_obs", -1),
                                         d.rnn_states_critic[-1].reshape(-1, d.hidden_size) if rec else None,
                                         d.masks[-1].reshape(-1, 1) if rec else None).view(
                d.n_rollout_threads, d.num_agents, 1)
        self.buffer.compute_returns(next_values, mod.get_critic_value_normalizer())

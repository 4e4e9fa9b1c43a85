"""Host-side logic added with the recurrent / MPE / A2C work - runs without a GPU: loud failures of the product
path, host queries of the new C entry points, recurrent parameter bookkeeping, example configs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from openrl_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


def test_recurrent_host_queries_match_the_reference_parameter_count():
    lib = nat.load()
    for D, K, head in ((18, 5, nat.ORL_HEAD_CATEGORICAL), (54, 1, nat.ORL_HEAD_VALUE), (6, 2, nat.ORL_HEAD_GAUSSIAN)):
        net = nat.NetDesc(D, 64, K, head)
        H = 64
        want = H * D + 3 * H + H * H + 3 * H + 2 * 3 * H * H + 2 * 3 * H + 2 * H + K * H + K
        want += K if head == nat.ORL_HEAD_GAUSSIAN else 0
        assert lib.orl_rnn_param_count(C.byref(net)) == want
        # same count as torch's own modules in the reference's registration order
        gru = torch.nn.GRU(H, H)
        n_ref = sum(p.numel() for p in gru.parameters()) + sum(
            p.numel() for m in (torch.nn.Linear(D, H), torch.nn.LayerNorm(H), torch.nn.Linear(H, H), torch.nn.LayerNorm(H),
                                torch.nn.LayerNorm(H), torch.nn.Linear(H, K)) for p in m.parameters())
        assert want - (K if head == nat.ORL_HEAD_GAUSSIAN else 0) == n_ref
        raw = lib.orl_rnn_raw_grad_count(C.byref(net))
        assert raw == H * D + H * H + 2 * 3 * H * H + K * H + 2 * H + 2 * 3 * H + K + (K if head == nat.ORL_HEAD_GAUSSIAN else 0)
    assert lib.orl_mpe_state_width() == 24
    pn, cn = nat.NetDesc(18, 64, 5, nat.ORL_HEAD_CATEGORICAL), nat.NetDesc(54, 64, 1, nat.ORL_HEAD_VALUE)
    ws = lib.orl_rnn_workspace_floats(C.byref(pn), C.byref(cn), 76800, 2)
    assert 200e6 < ws * 4 < 2e9  # two 442 MB tapes + state tapes + partial rows at the cfg4 shape


def test_recurrent_entry_points_validate_before_launching():
    lib = nat.load()
    net = nat.NetDesc(18, 64, 5, nat.ORL_HEAD_CATEGORICAL)
    assert lib.orl_rnn_chunk_rows(None, 0, 2, 25, 8, None, None) == -1
    assert b"orl_rnn_chunk_rows" in lib.orl_last_error_string()
    assert lib.orl_mpe_step(None, None, None, None, None, None, None, 4, 0, 25, None) == -1
    bad = nat.NetDesc(18, 128, 5, nat.ORL_HEAD_CATEGORICAL)
    assert lib.orl_rnn_ppo_apply(C.byref(bad), C.byref(net), None, None, None, None, None, None, None) == -2  # hidden 128
    assert b"hidden_size 128" in lib.orl_last_error_string()


@pytest.mark.skipif(not NO_GPU, reason="checks the loud failure on a box without a HIP device")
def test_new_product_paths_fail_loudly_without_a_gpu():
    from openrl_amd.envs.common import make

    with pytest.raises(nat.NativeError):
        make("simple_spread", env_num=4)
    with pytest.raises(NotImplementedError):
        make("HalfCheetah-v4", env_num=4)


def test_a2c_and_example_configs():
    from openrl_amd.algorithms.a2c import A2CAlgorithm
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.configs.config import create_config_parser
    from openrl_amd.runners.common import A2CAgent, PPOAgent

    assert issubclass(A2CAlgorithm, PPOAlgorithm) and issubclass(A2CAgent, PPOAgent)
    cfg = create_config_parser().parse_args(["--config", os.path.join(ROOT, "examples", "mpe", "mpe_ppo.yaml")])
    assert cfg.use_recurrent_policy is True and cfg.episode_length == 25 and cfg.lr == pytest.approx(7e-4)
    assert cfg.data_chunk_length == 2 and cfg.recurrent_N == 1  # reference defaults (config.py:578-589)


def test_recurrent_tower_state_dict_keys_follow_the_reference():
    from openrl_amd.modules.ppo_module import _tower_entries

    keys = [k for k, _ in _tower_entries("policy", 18, 64, 5, False, True)]
    assert keys[8:14] == ["rnn.rnn.weight_ih_l0", "rnn.rnn.weight_hh_l0", "rnn.rnn.bias_ih_l0", "rnn.rnn.bias_hh_l0",
                          "rnn.norm.weight", "rnn.norm.bias"]
    assert keys[-2:] == ["act.action_out.linear.weight", "act.action_out.linear.bias"]
    n = sum(int(np.prod(s)) for _, s in _tower_entries("critic", 54, 64, 1, False, True))
    lib = nat.load()
    assert n == lib.orl_rnn_param_count(C.byref(nat.NetDesc(54, 64, 1, nat.ORL_HEAD_VALUE)))


def test_device_train_info_is_a_dict_that_copies_on_first_read():
    """PPOAlgorithm.train returns its six averages without a device->host sync; any read materialises them."""
    from openrl_amd.algorithms.ppo import INFO_KEYS, DeviceTrainInfo

    vals = torch.tensor([0.5, -0.25, 1.5, 2.0, 3.0, 1.0])
    d = DeviceTrainInfo(INFO_KEYS, vals)
    assert isinstance(d, dict) and d._dev is not None
    assert d["value_loss"] == 0.5 and d._dev is None           # first read copies
    assert list(d.keys()) == list(INFO_KEYS) and len(d) == 6 and "ratio" in d
    assert dict(DeviceTrainInfo(INFO_KEYS, vals)) == {k: float(v) for k, v in zip(INFO_KEYS, vals)}
    e = DeviceTrainInfo(INFO_KEYS, vals)
    e.pop("ratio", None)                                        # A2CAlgorithm.train drops the ratio
    assert "ratio" not in e and len(e) == 5
    f = {}
    f.update(DeviceTrainInfo(INFO_KEYS, vals))
    assert f["dist_entropy"] == 1.5
    assert bool(DeviceTrainInfo(INFO_KEYS, vals)) and np.isfinite(list(DeviceTrainInfo(INFO_KEYS, vals).values())).all()


# ---- foreign callback objects (HipDriver injected into the reference's PPOAgent) ----------
class _ForeignBase:
    """Shape of the reference's ``openrl.utils.callbacks.callbacks.BaseCallback`` - NOT a subclass of this package's."""

    def __init__(self):
        self.agent, self.n_calls, self.locals, self.started = None, 0, {}, 0

    def init_callback(self, agent):
        self.agent = agent

    def on_rollout_start(self):
        self.started += 1

    def update_locals(self, locals_):
        self.locals.update(locals_)

    def on_step(self):
        self.n_calls += 1
        return self._on_step()

    def _on_step(self):
        return True

    def on_rollout_end(self):
        pass


def _foreign(name, **attrs):
    cls = type(name, (_ForeignBase,), {})
    obj = cls()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


def test_foreign_callback_objects_are_used_as_they_are():
    """rl_agent._init_callback (reference rl_agent.py:137-164) hands a driver its OWN CallbackList / ConvertCallback
    object even for ``callback=None``; it must be driven through its hooks, not wrapped and called as a function."""
    from openrl_amd.utils import callbacks as cb

    none_cb = _foreign("ConvertCallback", callback=None)          # what callback=None becomes in the reference
    got = cb.as_callback(none_cb)
    assert got is none_cb                                          # not wrapped into this package's ConvertCallback
    assert cb.callback_needs_per_step(none_cb) is False            # -> the fused rollout stays available
    fn_cb = _foreign("ConvertCallback", callback=lambda l, g: True)
    assert cb.as_callback(fn_cb) is fn_cb and cb.callback_needs_per_step(fn_cb) is True
    lst = _foreign("CallbackList", callbacks=[none_cb, _foreign("CallbackList", callbacks=[])])
    assert cb.as_callback(lst) is lst and cb.callback_needs_per_step(lst) is False
    lst2 = _foreign("CallbackList", callbacks=[none_cb, _foreign("CheckpointCallback")])
    assert cb.callback_needs_per_step(lst2) is True
    # a plain function is still converted, and a mixed list of ours + foreign objects is a CallbackList of ours
    conv = cb.as_callback(lambda l, g: False)
    assert isinstance(conv, cb.ConvertCallback) and conv.needs_per_step
    mixed = cb.CallbackList([cb.NoopCallback(), none_cb])
    assert mixed.needs_per_step is False
    # the hooks are called on the foreign object itself
    got.init_callback("agent")
    got.on_rollout_start()
    got.update_locals({"obs": 1})
    assert got.on_step() is True and got.n_calls == 1 and got.locals["obs"] == 1 and got.started == 1


def test_stop_training_on_no_model_improvement_counts_like_the_reference():
    """stop_callback.py:107-154: improvements reset the counter, the (max+1)-th stale evaluation stops training,
    evaluations up to ``min_evals`` are not counted."""
    from openrl_amd.utils import callbacks as cb

    class _Parent:
        best_mean_reward = -float("inf")

    c = cb.CallbackFactory.get_callback({"id": "StopTrainingOnNoModelImprovement",
                                         "args": {"max_no_improvement_evals": 2, "min_evals": 1, "verbose": 0}})
    c.parent = _Parent()
    c.init_callback(type("A", (), {"num_time_steps": 0})())
    seq = [1.0, 1.0, 2.0, 2.0, 2.0, 2.0]   # best_mean_reward after each evaluation
    out = []
    for r in seq:
        c.parent.best_mean_reward = r
        out.append(c.on_step())
    # call 1 (<= min_evals) not counted; call 2 stale (1); call 3 improves; calls 4, 5 stale (1, 2); call 6 stale (3) -> stop
    assert out == [True, True, True, True, True, False]


def test_callback_tree_follows_the_reference_constructors():
    """callbacks.py:133-308, eval_callback.py:81-146, callbacks_factory.py:28-66: children given as objects or as
    ``{"id": ...}`` specs, ``set_parent`` through lists, OR / AND stop logic (every child runs either way), the
    reference's import paths."""
    from openrl_amd.utils.callbacks.callbacks import BaseCallback, CallbackList, EveryNTimesteps
    from openrl_amd.utils.callbacks.callbacks_factory import CallbackFactory
    from openrl_amd.utils.callbacks.checkpoint_callback import CheckpointCallback
    from openrl_amd.utils.callbacks.eval_callback import EvalCallback
    from openrl_amd.utils.callbacks.stop_callback import StopTrainingOnRewardThreshold

    class _Ret(BaseCallback):
        def __init__(self, ret):
            super().__init__()
            self.ret = ret

        def _on_step(self):
            return self.ret

    agent = type("A", (), {"num_time_steps": 0, "get_env": lambda self: "env", "logger": "lg"})()
    for logic, rets, want in (("OR", [True, False, True], False), ("OR", [True, True], True),
                              ("AND", [True, False], True), ("AND", [False, False], False)):
        kids = [_Ret(r) for r in rets]
        lst = CallbackList(kids, stop_logic=logic)
        lst.init_callback(agent)
        assert lst.on_step() is want
        assert all(k.n_calls == 1 and k.training_env == "env" and k.logger == "lg" for k in kids)
    with pytest.raises(ValueError):
        CallbackList([], stop_logic="XOR")

    ev = EvalCallback({"id": "CartPole-v1", "env_num": 2},
                      callbacks_on_new_best=[{"id": "StopTrainingOnRewardThreshold", "args": {"reward_threshold": 5}}],
                      callbacks_after_eval=StopTrainingOnRewardThreshold(7), stop_logic="AND", log_path="/tmp/x",
                      warn=False, render=False, asynchronous=False, close_env_at_end=False)
    assert ev.callbacks_on_new_best.stop_logic == "AND" and ev.callbacks_on_new_best.callbacks[0].parent is ev
    assert ev.callback.parent is ev and ev.log_path == os.path.join("/tmp/x", "evaluations")
    every = EveryNTimesteps(n_steps=6, callbacks={"id": "CheckpointCallback", "args": {"save_freq": 1, "save_path": "/tmp/x"}})
    assert isinstance(every.callback.callbacks[0], CheckpointCallback) and every.callback.callbacks[0].parent is every
    CallbackFactory.register("Ret", _Ret)
    assert isinstance(CallbackFactory.get_callback({"id": "Ret", "args": {"ret": True}}), _Ret)


def test_logger_run_dirs_and_file_scalar_backend(tmp_path):
    """logger.py:64-151,185-207: <log_path>/<project>/<scenario>/<exp>/run<k>, log.txt, and every log_info key reaching
    the scalar back-end (tensorboardX when importable, else the scalars.jsonl / scalars.csv writer)."""
    import json

    from openrl_amd.configs.config import default_cfg
    from openrl_amd.utils.logger import Logger

    cfg = default_cfg([])
    for k in (1, 2):
        lg = Logger(cfg, project_name="proj", scenario_name="scen", exp_name="exp", log_path=str(tmp_path),
                    use_tensorboard=True)
        assert lg.run_dir == tmp_path / "proj" / "scen" / "exp" / ("run%d" % k)
        lg.log_info({"value_loss": torch.tensor(0.5), "FPS": 123, "rewards": [1.0, 3.0]}, step=64 * k)
        lg.log_learner_info(0, {"lr": 5e-4}, step=64 * k)
        lg.info("hello")
        lg.close()
        assert (lg.run_dir / "log.txt").exists() and "value_loss: 0.5" in (lg.run_dir / "log.txt").read_text()
        assert lg.history[-1] == (64 * k, {"value_loss": 0.5, "FPS": 123.0, "rewards": 2.0})
        jl = lg.run_dir / "logs" / "scalars.jsonl"
        if jl.exists():  # tensorboardX absent: the file back-end
            rows = [json.loads(l) for l in jl.read_text().splitlines()]
            assert {r["key"] for r in rows} == {"value_loss", "FPS", "rewards", "Learner_0/lr"}
            assert all(r["step"] == 64 * k for r in rows)
    quiet = Logger(cfg)  # no log_path: nothing on disk, history only
    quiet.log_info({"a": 1}, 1)
    assert quiet.run_dir is None and quiet.history == [(1, {"a": 1.0})]


# ---- tests/test_buffer/test_buffer.py:28-55 with the import swapped -------------------------------------------------
def test_obs_data_basic():
    from openrl_amd.buffers.utils.obs_data import ObsData

    a_data = np.array([1, 2])
    b_data = np.array([3, 4])
    c_data = np.array([5, 6, 7])
    obs = ObsData({"a": a_data, "b": b_data, "c": c_data})

    assert np.all(obs["c"] == c_data)
    obs["a"][0] = 99
    a_data[0] = 99
    assert np.all(obs["a"] == a_data)

    assert np.all(obs.flatten() == np.concatenate([a_data, b_data, c_data]))


def test_obs_data_step():
    from openrl_amd.buffers.utils.obs_data import ObsData

    obs_stepes = [
        {"obs_a": np.array([[[0, 1]]]), "obs_b": np.array([[[3, 9]]])},
        {"obs_a": np.array([[[2, 4]]]), "obs_b": np.array([[[6, 8]]])},
    ]
    obs = ObsData({"obs_a": np.zeros((2, 1, 1, 2)), "obs_b": np.zeros((2, 1, 1, 2))})
    for step in range(len(obs_stepes)):
        for key in obs.keys():
            obs[key][step] = obs_stepes[step][key]

    step_data = {"obs_a": np.array([[0.0, 1.0]]), "obs_b": np.array([[3.0, 9.0]])}
    for key in obs[0]:
        assert np.all(obs[0][key] == step_data[key])
    assert obs.all_batch(0, 2)["obs_b"].shape == (2, 2) and np.all(obs.all_batch(1, 2)["obs_a"] == [[2.0, 4.0]])
    assert np.all(ObsData.prepare_input({"p": np.zeros((3, 2, 5))})["p"].shape == (6, 5))

"""The reference's example scripts with the imports swapped (examples/), shortened: train, save, load, evaluate."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    spec = importlib.util.spec_from_file_location("example_" + os.path.basename(os.path.dirname(path)), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mpe_example_train_save_load_evaluate(tmp_path):
    ex = _load(os.path.join(ROOT, "examples", "mpe", "train_ppo.py"))
    save_dir = str(tmp_path / "ppo_agent")
    agent = ex.train(env_num=64, total_time_steps=64 * 25 * 2, save_dir=save_dir,
                     argv=["--config", os.path.join(ROOT, "examples", "mpe", "mpe_ppo.yaml"), "--ppo_epoch", "2"])
    assert agent.net.module.recurrent and os.path.exists(os.path.join(save_dir, "module.pt"))
    total_reward, steps = ex.evaluation(agent, env_num=9, save_dir=save_dir)
    assert steps == 25 and np.isfinite(total_reward) and total_reward < 0


def test_mpe_jrpo_and_mat_examples(tmp_path):
    ex = _load(os.path.join(ROOT, "examples", "mpe", "train_ppo.py"))
    agent = ex.train(env_num=32, total_time_steps=32 * 25 * 2, save_dir=str(tmp_path / "jrpo_agent"),
                     argv=["--config", os.path.join(ROOT, "examples", "mpe", "mpe_jrpo.yaml"), "--ppo_epoch", "2"])
    assert agent.driver.trainer.use_joint_action_loss and agent.net.module.recurrent
    mat = _load(os.path.join(ROOT, "examples", "mpe", "train_mat.py"))
    agent = mat.train(env_num=32, total_time_steps=32 * 25 * 2,
                      argv=["--config", os.path.join(ROOT, "examples", "mpe", "mpe_mat.yaml"), "--ppo_epoch", "2"])
    assert agent.driver.trainer.__class__.__name__ == "MATAlgorithm" and not agent.net.module.recurrent
    total_reward, steps = mat.evaluation(agent, env_num=9)
    assert steps == 25 and np.isfinite(total_reward) and total_reward < 0


def test_cartpole_example_train_and_evaluate():
    ex = _load(os.path.join(ROOT, "examples", "cartpole", "train_ppo.py"))
    agent = ex.train(env_num=9, total_time_steps=9 * 200 * 2)
    steps, total_reward = ex.evaluation(agent, env_num=9)
    assert steps >= 8 and total_reward > 0


def test_selfplay_and_callback_examples_run(tmp_path, monkeypatch, capsys):
    import sys

    monkeypatch.chdir(tmp_path)  # the callbacks example writes ./results/
    sp = _load(os.path.join(ROOT, "examples", "selfplay", "train_selfplay.py"))
    monkeypatch.setattr(sys, "argv", ["train_selfplay.py", "--envs", "512", "--steps", str(512 * 20 * 30)])
    sp.main()
    out = capsys.readouterr().out
    assert "vs random opponent" in out and float(out.strip().split()[-1]) > 0.2
    cb = _load(os.path.join(ROOT, "examples", "cartpole", "train_ppo_callbacks.py"))
    cb.main()
    out = capsys.readouterr().out
    assert "stopped after" in out and os.path.isdir(tmp_path / "results" / "best_model")

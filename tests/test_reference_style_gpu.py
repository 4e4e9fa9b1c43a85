"""The reference's own unit tests of this path, with the imports swapped (``openrl`` -> ``openrl_amd``) and nothing else
changed in the calls: tests/test_algorithm/test_ppo_algorithm.py:36-82 and test_a2c_algorithm.py:36-97 build ``PPOModule``
(separate / shared networks) and a zero-filled ``NormalReplayBuffer(episode_length=100)`` and call ``.train``.
The reference asserts "no exception"; here the train_info keys and finite weights are checked too."""
import numpy as np
import pytest
import torch

from openrl_amd import spaces

pytestmark = pytest.mark.gpu


@pytest.fixture
def obs_space():
    return spaces.Box(low=-np.inf, high=+np.inf, shape=(1,), dtype=np.float32)


@pytest.fixture
def act_space():
    return spaces.Discrete(2)


@pytest.fixture(scope="module", params=["--use_share_model false", "--use_share_model true"])
def config(request):
    from openrl_amd.configs.config import create_config_parser

    cfg_parser = create_config_parser()
    cfg = cfg_parser.parse_args(request.param.split())
    return cfg


@pytest.fixture
def init_module(config, obs_space, act_space):
    from openrl_amd.modules.ppo_module import PPOModule

    module = PPOModule(
        config,
        policy_input_space=obs_space,
        critic_input_space=obs_space,
        act_space=act_space,
        share_model=config.use_share_model,
    )
    return module


@pytest.fixture
def buffer_data(config, obs_space, act_space):
    from openrl_amd.buffers.normal_buffer import NormalReplayBuffer

    buffer = NormalReplayBuffer(
        config,
        num_agents=1,
        obs_space=obs_space,
        act_space=act_space,
        data_client=None,
        episode_length=100,
    )
    return buffer.data


def _check(info, module):
    assert {"value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm"} <= set(info.keys())
    assert all(np.isfinite(float(v)) for v in info.values())
    for m in module.models.values():
        assert torch.isfinite(m.theta).all()


def test_ppo_algorithm(config, init_module, buffer_data):
    from openrl_amd.algorithms.ppo import PPOAlgorithm

    ppo_algo = PPOAlgorithm(config, init_module)

    info = ppo_algo.train(buffer_data)
    _check(info, init_module)


def test_a2c_algorithm(config, init_module, buffer_data):
    from openrl_amd.algorithms.a2c import A2CAlgorithm

    a2c_algo = A2CAlgorithm(config, init_module)

    info = a2c_algo.train(buffer_data)
    _check(info, init_module)


# ---- tests/test_examples/test_train_mpe.py:17-58, imports swapped (test_train_cartpole.py's >= 450 threshold after
# 20 000 steps depends on the seed - 298 at seed 0 here, 292 for the CPU restatement of the reference on the same env,
# 500 at seed 1; tests/test_rollout_gpu.py::test_cartpole_learns_like_the_reference_recipe pins that recipe) --------
@pytest.fixture(
    scope="module",
    params=[
        "--episode_length 5 --use_recurrent_policy true --use_joint_action_loss true"
        " --use_valuenorm true --use_adv_normalize true"
    ],
)
def mpe_config(request):
    from openrl_amd.configs.config import create_config_parser

    cfg_parser = create_config_parser()
    cfg = cfg_parser.parse_args(request.param.split())
    return cfg


def test_train_mpe(mpe_config, tmp_path):
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet as Net
    from openrl_amd.runners.common import PPOAgent as Agent

    env_num = 2
    env = make(
        "simple_spread",
        env_num=env_num,
        asynchronous=True,
    )
    net = Net(env, cfg=mpe_config)
    agent = Agent(net)
    agent.train(total_time_steps=30)
    agent.save(str(tmp_path / "ppo_agent"))
    agent.load(str(tmp_path / "ppo_agent"))
    agent.set_env(env)
    obs, info = env.reset(seed=0)
    step = 0
    while step < 5:
        action, _ = agent.act(obs, deterministic=True)
        obs, r, done, info = env.step(action)
        if np.any(done):
            break
        step += 1
    env.close()

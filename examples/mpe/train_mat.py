"""MPE simple_spread with ``MATAgent`` - the reference's examples/mpe/train_mat.py with the imports swapped, except that
the network is the MLP ``PPONet`` (the reference's ``MATNet`` transformer is not built here):
python examples/mpe/train_mat.py --config examples/mpe/mpe_mat.yaml"""
import numpy as np

from openrl_amd.configs.config import create_config_parser
from openrl_amd.envs.common import make
from openrl_amd.modules.common import PPONet as Net
from openrl_amd.runners.common import MATAgent as Agent


def train(env_num=2048, total_time_steps=5000000, argv=None):
    env = make("simple_spread", env_num=env_num, asynchronous=True)
    cfg = create_config_parser().parse_args(argv)
    net = Net(env, cfg=cfg, device="cuda")
    agent = Agent(net, use_wandb=False)
    agent.train(total_time_steps=total_time_steps)
    env.close()
    return agent


def evaluation(agent, env_num=9):
    env = make("simple_spread", env_num=env_num, asynchronous=False)
    agent.set_env(env)
    obs, info = env.reset(seed=0)
    done, step, total_reward = False, 0, 0
    while not np.any(done):
        action, _ = agent.act(obs, deterministic=True)
        obs, r, done, info = env.step(action)
        step += 1
        total_reward += np.mean(r)
    print(f"total_reward: {total_reward}")
    env.close()
    return total_reward, step


if __name__ == "__main__":
    evaluation(train())

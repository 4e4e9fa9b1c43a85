"""CartPole-v1 PPO - the reference's examples/cartpole/train_ppo.py with the three imports swapped."""
import numpy as np

from openrl_amd.envs.common import make
from openrl_amd.modules.common import PPONet as Net
from openrl_amd.runners.common import PPOAgent as Agent


def train(env_num=9, total_time_steps=20000):
    env = make("CartPole-v1", env_num=env_num)
    net = Net(env, device="cuda")
    agent = Agent(net)
    agent.train(total_time_steps=total_time_steps)
    env.close()
    return agent


def evaluation(agent, env_num=9):
    env = make("CartPole-v1", env_num=env_num)
    agent.set_env(env)
    obs, info = env.reset()
    done, step, total_reward = False, 0, 0
    while not np.any(done):
        action, _ = agent.act(obs, deterministic=True)
        obs, r, done, info = env.step(action)
        step += 1
        total_reward += np.mean(r)
    print(f"total steps: {step}, total reward: {total_reward}")
    env.close()
    return step, total_reward


if __name__ == "__main__":
    evaluation(train())

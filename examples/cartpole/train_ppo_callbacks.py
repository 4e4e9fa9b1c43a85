"""CartPole PPO with the callbacks taken from a yaml file (cfg.callbacks -> CallbackFactory): periodic checkpoints, an
evaluation env on the device, stop at an evaluation return of 400.

    python examples/cartpole/train_ppo_callbacks.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from openrl_amd.configs.config import create_config_parser  # noqa: E402
from openrl_amd.envs.common import make  # noqa: E402
from openrl_amd.modules.common import PPONet as Net  # noqa: E402
from openrl_amd.runners.common import PPOAgent as Agent  # noqa: E402
from openrl_amd.utils.callbacks import CallbackFactory  # noqa: E402


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = create_config_parser().parse_args(["--config", os.path.join(here, "callbacks.yaml")])
    env = make("CartPole-v1", env_num=256)
    agent = Agent(Net(env, cfg=cfg, device="cuda:0"))
    callbacks = CallbackFactory.get_callbacks(cfg.callbacks)
    agent.train(total_time_steps=256 * 64 * 2000, callback=callbacks)
    ev = [c for c in callbacks.callbacks if hasattr(c, "evaluations")][0]
    print("stopped after %d steps; evaluations (steps, mean return, episodes):" % agent.num_time_steps)
    for e in ev.evaluations:
        print("  %9d  %7.1f  %d" % e)


if __name__ == "__main__":
    main()

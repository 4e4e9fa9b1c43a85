"""Self-play PPO on tic-tac-toe (the reference's examples/selfplay/train_selfplay.py) on the MI355X engine: 4096
device-resident games, the opponent of env group g is snapshot g of a pool of frozen copies of the learner, refreshed
every 10 rollouts by SelfPlayCallback; legal-move masks never leave the device.

    python examples/selfplay/train_selfplay.py [--envs 4096 --steps 2000000]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from openrl_amd.configs.config import create_config_parser  # noqa: E402
from openrl_amd.envs.common import make  # noqa: E402
from openrl_amd.modules.common import PPONet as Net  # noqa: E402
from openrl_amd.runners.common import PPOAgent as Agent  # noqa: E402
from openrl_amd.utils.callbacks import SelfPlayCallback  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2_000_000)
    a = ap.parse_args()
    cfg = create_config_parser().parse_args(["--config", os.path.join(os.path.dirname(__file__), "selfplay.yaml")])
    env = make("tictactoe_v3", env_num=a.envs, opponent="pool", pool_size=4, device="cuda:0")
    agent = Agent(Net(env, cfg=cfg, device="cuda:0"))
    agent.train(total_time_steps=a.steps, callback=SelfPlayCallback(push_every=10))
    # evaluation against the uniformly random opponent (RandomOpponentWrapper of the reference's evaluation())
    ev = make("tictactoe_v3", env_num=4096, device="cuda:0", seed=7)
    obs = ev.reset_device(seed=7)
    for _ in range(50):
        act, _ = agent.net.module.act(obs.view(4096, 18), None, None, action_masks=ev.action_mask_device.view(4096, 9),
                                      deterministic=True)
        obs, _, _ = ev.step_device(act.view(4096, 1, 1))
    st = ev.episode_statistics()
    print("vs random opponent: %d games, mean result %.3f" % (st["episodes_finished"], st["episode_return_mean"]))


if __name__ == "__main__":
    main()

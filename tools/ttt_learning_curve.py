"""Learning evidence for config 5's env: mean game result against the uniformly random opponent, evaluated every few
iterations, for (a) PPO trained against the random opponent and (b) PPO trained by self-play against the snapshot pool.

    python tools/ttt_learning_curve.py > profiles/r01_ttt_learning.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def evaluate(module, seed=99, n=4096, steps=40, dev="cuda:0"):
    from openrl_amd.envs.common import make

    ev = make("tictactoe_v3", env_num=n, device=dev, seed=seed)
    obs = ev.reset_device(seed=seed)
    for _ in range(steps):
        a, _ = module.act(obs.view(n, 18), None, None, action_masks=ev.action_mask_device.view(n, 9), deterministic=True)
        obs, _, _ = ev.step_device(a.view(n, 1, 1))
    st = ev.episode_statistics()
    return round(st["episode_return_mean"], 4)


def run(opponent, iters=200, every=20, N=1024, T=10, dev="cuda:0"):
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    cfg = default_cfg(["--seed", "0", "--lr", "1e-3", "--critic_lr", "1e-3", "--episode_length", str(T), "--ppo_epoch", "5",
                       "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = make("tictactoe_v3", env_num=N, device=dev, opponent=opponent, pool_size=4)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * iters

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    curve = [(0, evaluate(net.module))]
    for i in range(iters):
        if opponent == "pool" and i and i % 10 == 0:
            env.push_opponent(net.module.models["policy"].theta)
        drv.episode = i
        drv._inner_loop()
        if (i + 1) % every == 0:
            curve.append(((i + 1) * N * T, evaluate(net.module)))
    return curve


if __name__ == "__main__":
    out = {"metric": "mean game result vs the uniformly random opponent (greedy policy, 4096 games x 40 steps)",
           "trained_vs_random_opponent": run("random"), "trained_by_self_play_vs_snapshot_pool": run("pool")}
    print(json.dumps(out, indent=1))

"""How far ahead of the GPU does the host run?  Enqueue time per training iteration (no synchronisation) against the
synchronised time of the same iterations, configuration 2's loop at a given env count.  Usage: host_rate_probe.py [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.normal_buffer import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = 40
    dev = torch.device("cuda:0")
    cfg = default_cfg(["--episode_length", "128", "--ppo_epoch", "10", "--amd_perm_mode", "device", "--log_interval",
                       "1000000"])
    env = make("SyntheticFixedStep-v0", env_num=n, obs_dim=4, episode_limit=200, device=dev, seed=1)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=n)
    cfg.num_env_steps = n * 128 * (iters * 3 + 16)
    algo = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)

    class _A:
        num_time_steps = 0

    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, algo, buf, _A(),
                         rank=0, world_size=1)
    drv.reset_and_buffer_init()
    ep = 0
    for _ in range(5):
        drv.episode = ep
        ep += 1
        drv._inner_loop()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(iters):
            drv.episode = ep
            ep += 1
            drv._inner_loop()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("envs %d: host enqueue %.1f us per iteration, synchronised %.1f us per iteration" %
              (n, (t1 - t0) / iters * 1e6, (t2 - t0) / iters * 1e6))


if __name__ == "__main__":
    main()

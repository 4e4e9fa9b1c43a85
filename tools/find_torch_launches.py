"""Which torch (aten) ops does one training iteration of the bench workload still launch, and from where?
    python tools/find_torch_launches.py [--envs 4096]
Runs 3 iterations of bench.py's loop under torch.profiler (CPU activity, python stacks) and prints every aten op that
launches a device kernel (fill / zero / mul / copy / empty are the usual suspects) with its innermost repo frame."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    a = ap.parse_args()
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    dev, N, T = "cuda:0", a.envs, 128
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "10", "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = make("SyntheticFixedStep-v0", env_num=N, obs_dim=4, episode_limit=200, device=dev)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * 100

    class _A:
        num_time_steps = 0

    tr = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, tr, buf, _A())
    drv.reset_and_buffer_init()
    for i in range(3):
        drv.episode = i
        drv._inner_loop()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        for i in range(3):
            drv.episode = 3 + i
            drv._inner_loop()
        torch.cuda.synchronize()
    seen = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        frame = next((f for f in (ev.stack or []) if "openrl_amd" in f or "bench" in f), "?")
        key = (ev.name, frame)
        seen[key] = seen.get(key, 0) + 1
    for (name, frame), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("%3d x %-28s %s" % (n, name, frame))


if __name__ == "__main__":
    main()

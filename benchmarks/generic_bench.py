"""The general tower path at BASELINE config 2's shape with a non-default tower, e.g.
--hidden_size 128 --layer_N 2: env-steps/s of collect (stepwise) + GAE + PPO update.  Not the bench.py headline.

    python benchmarks/generic_bench.py [--hidden_size 128 --layer_N 1 --envs 4096 --steps 3 --warmup 1]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--hidden_size", type=int, default=128)
    ap.add_argument("--layer_N", type=int, default=1)
    ap.add_argument("--share", action="store_true")
    ap.add_argument("--gen_update", default="fused", choices=["fused", "layerwise"])
    ap.add_argument("--activation_id", type=int, default=1)
    ap.add_argument("--obs_dim", type=int, default=4)
    ap.add_argument("--one_stream", action="store_true", help="both towers' update chains on one stream (per-kernel timing)")
    a = ap.parse_args()
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    dev, N, T = "cuda:0", a.envs, a.T
    argv = ["--episode_length", str(T), "--ppo_epoch", "10", "--amd_perm_mode", "device", "--log_interval", "1000000",
            "--hidden_size", str(a.hidden_size), "--layer_N", str(a.layer_N), "--amd_gen_update", a.gen_update,
            "--activation_id", str(a.activation_id)]
    if a.share:
        argv += ["--use_share_model", "true"]
    cfg = default_cfg(argv)
    env = make("SyntheticFixedStep-v0", env_num=N, obs_dim=a.obs_dim, episode_limit=200, device=dev)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * (a.steps + a.warmup)
    if a.one_stream and getattr(net.module, "generic", False):
        net.module.two_stream = False

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    for i in range(a.warmup):
        drv.episode = i
        drv._inner_loop()
    torch.cuda.synchronize()
    t_roll = t_upd = 0.0
    t0 = time.perf_counter()
    for i in range(a.steps):
        drv.episode = a.warmup + i
        ta = time.perf_counter()
        drv.actor_rollout()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        drv.learner_update()
        drv.buffer.after_update()
        torch.cuda.synchronize()
        t_roll += tb - ta
        t_upd += time.perf_counter() - tb
    dt = time.perf_counter() - t0
    print(json.dumps({"bench": "generic_tower_path", "envs": N, "rollout_len": T, "hidden_size": a.hidden_size,
                      "layer_N": a.layer_N, "share_model": a.share, "gen_update": a.gen_update, "generic": bool(getattr(net.module, "generic", False)),
                      "env_steps_per_s": N * T * a.steps / dt, "ms_per_iteration": dt / a.steps * 1e3,
                      "rollout": "fused" if drv.fused else "graph" if drv._graph is not None else "stepwise",
                      "ms_rollout": t_roll / a.steps * 1e3, "ms_update": t_upd / a.steps * 1e3}))


if __name__ == "__main__":
    main()

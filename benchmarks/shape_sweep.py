"""env-steps/s (collect + GAE + PPO update) of the MLP path at the SHAPES of BASELINE.json's other single-agent
configs, on the synthetic fixed-step env: config 3 (HalfCheetah shape: 1024 envs, obs 17,
Box(6) Gaussian policy, 200-step rollout) and config 5 (tictactoe shape: 4096 envs, obs 18, Discrete(9), 200 steps;
the synthetic env's action masks are all ones).  Reference defaults otherwise (ppo_epoch 10, num_mini_batch 1).

    python benchmarks/shape_sweep.py [--steps 10 --warmup 2]

One JSON line per shape.  These are measurements beside the headline, not bench.py's metric."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {
    "cfg2_cartpole_shape": dict(envs=4096, T=128, obs=4, act=("discrete", 2)),
    "cfg3_halfcheetah_shape": dict(envs=1024, T=200, obs=17, act=("box", 6)),
    "cfg5_tictactoe_shape": dict(envs=4096, T=200, obs=18, act=("discrete", 9)),
}


def run(name, shp, steps, warmup, dev="cuda:0", tower_gemm="split"):
    import torch

    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T = shp["envs"], shp["T"]
    kind, n = shp["act"]
    act = spaces.Discrete(n) if kind == "discrete" else spaces.Box(-1.0, 1.0, (n,))
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "10", "--amd_perm_mode", "device",
                       "--log_interval", "1000000", "--amd_tower_gemm", tower_gemm])
    env = make("SyntheticFixedStep-v0", env_num=N, obs_dim=shp["obs"], action_space=act, episode_limit=200, device=dev)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * (steps + warmup + 1)

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, trainer, buf, _Agent())
    drv.reset_and_buffer_init()
    for i in range(warmup):
        drv.episode = i
        drv._inner_loop()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        drv.episode = warmup + i
        drv._inner_loop()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the dominant kernel's time comes from ONE more iteration with HIP events around its launches, outside the timed region:
    # 20 event records per iteration are ~80 us of marker packets on the launch stream (3.5 % of a 2.3 ms iteration)
    trainer.profile_events = []
    drv.episode = warmup + steps
    drv._inner_loop()
    torch.cuda.synchronize()
    ev = trainer.profile_events
    trainer.profile_events = None
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    flop_fwd = 2 * (2 * shp["obs"] * 64 + 2 * 64 * 64 + 64 * (n + 1))
    out = {"bench": name, "envs": N, "rollout_len": T, "obs_dim": shp["obs"], "action_space": "%s(%d)" % (kind, n),
           "fused_rollout": bool(drv.fused), "tower_gemm": tower_gemm, "tower_pair_ms": round(k_ms, 4),
           "tower_pair_frac_of_fp32_mfma_peak": round(3 * flop_fwd * N * T / (k_ms * 1e-3) / 157.3e12, 4) if k_ms else None,
           "ms_per_iteration": round(1e3 * dt / steps, 4),
           "env_steps_per_s": round(N * T * steps / dt, 1),
           "update_tflops_algorithmic": round(3 * flop_fwd * N * T * 10 * steps / dt / 1e12, 2)}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default=None)
    ap.add_argument("--tower-gemm", default="split", choices=["split", "fp32", "split_two_image"],
                    help="split_two_image = round 3's variants (no transposing-read full split): the A/B switch")
    a = ap.parse_args()
    for name, shp in SHAPES.items():
        if a.only and a.only not in name:
            continue
        run(name, shp, a.steps, a.warmup, tower_gemm=a.tower_gemm)


if __name__ == "__main__":
    main()

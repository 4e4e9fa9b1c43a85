"""Micro-benchmark of the recurrent (GRU) PPO update at the cfg4 shape of SURVEY.md section 8
(MPE simple_spread: N=2048 envs x 3 agents, T=25, obs 18 / 54, Discrete(5), data_chunk_length 2, ppo_epoch 10).

    python benchmarks/rnn_update_bench.py [--iters 5] [--chunk 2] [--envs 2048]

Prints one JSON line: ms per PPOAlgorithm.train call, rows/s, and the HIP-event time of the update kernels.
Random buffer contents (synthetic), weights random-init; measures the update only (the rollout of a recurrent
policy is the stepwise path)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=2)
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--T", type=int, default=25)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--tower-gemm", default="fp32", choices=["split", "fp32", "fp32_recompute", "split_w4"],
                    help="cfg.amd_rnn_gemm: fp32 (default) = every GEMM on v_mfma_f32_16x16x4_f32 out of resident LDS images; "
                         "split / split_w4 = the streamed bf16-split row kernel (8 / 4 waves per workgroup)")
    a = ap.parse_args()
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.ppo_module import PPOModule

    dev = "cuda:0"
    N, A, T = a.envs, a.agents, a.T
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", str(a.epochs), "--use_recurrent_policy", "true",
                       "--data_chunk_length", str(a.chunk), "--amd_perm_mode", "device",
                       "--amd_rnn_gemm", a.tower_gemm])
    cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = N, A, cfg.hidden_size
    box = lambda d: spaces.Box(-np.inf, np.inf, (d,))
    obs_space = spaces.Dict({"policy": box(18), "critic": box(54)})
    act_space = spaces.Discrete(5)
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=dev, rank=0, world_size=1)
    buf = ReplayData(cfg, A, obs_space, act_space, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    buf.policy_obs.copy_(r(T + 1, N, A, 18))
    buf.critic_obs.copy_(r(T + 1, N, A, 54))
    buf.rnn_states.copy_(0.3 * r(T + 1, N, A, 1, 64))
    buf.rnn_states_critic.copy_(0.3 * r(T + 1, N, A, 1, 64))
    buf.rewards.copy_(torch.rand(T, N, A, 1, device=dev, generator=g))
    buf.value_preds.copy_(0.3 * r(T + 1, N, A, 1))
    buf.masks.copy_((torch.rand(T + 1, N, A, 1, device=dev, generator=g) > 0.04).float())
    buf.actions.copy_(torch.randint(0, 5, (T, N, A, 1), device=dev, generator=g).float())
    buf.action_log_probs.copy_(np.log(0.2) + 0.05 * r(T, N, A, 1))
    algo = PPOAlgorithm(cfg, module, agent_num=A, device=dev)

    def one():
        buf.compute_returns(0.3 * r(N, A, 1), module.get_critic_value_normalizer())
        return algo.train(buf)

    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        info = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    M = T * N * A
    print(json.dumps({"bench": "rnn_update", "shape": {"envs": N, "agents": A, "T": T, "chunk": a.chunk,
                                                        "ppo_epoch": a.epochs, "rows": M},
                      "ms_per_train": dt * 1e3, "ms_per_epoch": dt * 1e3 / a.epochs,
                      "row_updates_per_s": M * a.epochs / dt, "info": {k: round(float(v), 5) for k, v in info.items()}}))


if __name__ == "__main__":
    main()

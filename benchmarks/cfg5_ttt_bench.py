"""BASELINE config 5's env end to end on one MI355X: tic-tac-toe against the uniformly random opponent
(examples/selfplay with RandomOpponentWrapper), 4096 envs, obs 18, Discrete(9) with legal-move masks that stay on the
device, episode_length 200 (the reference default), MLP PPO with the reference defaults; stepwise rollout
(orl_act_step with masks + orl_ttt_step + orl_buffer_insert per step, replayed as one hipGraph) + GAE + PPO update.

    python benchmarks/cfg5_ttt_bench.py [--steps 10 --warmup 2 --envs 4096 --T 200]

One JSON line: env-steps/s (N*T per iteration / wall) and the split rollout / update in ms.  The opponent-pool part of
config 5 (opponents = earlier checkpoints) is not built; an extra measurement beside bench.py's headline."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--opponent", default="random", choices=["random", "pool"],
                    help="pool: self-play against 4 frozen snapshots of the learner (stepwise rollout, hipGraph)")
    ap.add_argument("--sampling", default="per_rollout", choices=["per_reset", "per_rollout", "static"],
                    help="pool opponent assignment: per_reset = the reference's per-episode draw (stepwise rollout as a "
                         "hipGraph, per-env pool act launch); per_rollout / static keep one snapshot per 16-env tile, "
                         "which the fused rollout kernel plays in-kernel")
    ap.add_argument("--rollout-kernel", default="chain", choices=["chain", "lockstep"],
                    help="random opponent: the round-6 chain kernel or the round-5 lock-step kernel (A/B)")
    a = ap.parse_args()
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    dev, N, T = "cuda:0", a.envs, a.T
    cfg = default_cfg(["--seed", "0", "--episode_length", str(T), "--amd_perm_mode", "device", "--log_interval", "1000000",
                       "--amd_rollout_kernel", a.rollout_kernel])
    kw = dict(opponent_sampling=a.sampling) if a.opponent == "pool" else {}
    env = make("tictactoe_v3", env_num=N, device=dev, opponent=a.opponent, **kw)
    if a.opponent == "pool":
        torch.manual_seed(1)
        env.opp_thetas.copy_(0.1 * torch.randn_like(env.opp_thetas))  # non-trivial opponents from the start
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * (a.steps + a.warmup)

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
        tc = time.perf_counter()
        t_roll += tb - ta
        t_upd += tc - tb
    dt = time.perf_counter() - t0
    st = env.episode_statistics()
    print(json.dumps({"bench": "cfg5_tictactoe_%s_opponent" % a.opponent, "envs": N, "episode_length": T, "ppo_epoch": cfg.ppo_epoch,
                      "env_steps_per_s": N * T * a.steps / dt, "ms_per_iteration": dt / a.steps * 1e3,
                      "ms_rollout": t_roll / a.steps * 1e3, "ms_update": t_upd / a.steps * 1e3,
                      "games_finished": st["episodes_finished"], "mean_game_result": st["episode_return_mean"],
                      "rollout": "fused" if drv.fused else "stepwise (hipGraph)",
                      "opponent_sampling": a.sampling if a.opponent == "pool" else None}))


if __name__ == "__main__":
    main()

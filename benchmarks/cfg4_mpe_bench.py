"""BASELINE config 4 end to end on one MI355X: MPE simple_spread, 2048 envs x 3 agents, episode_length 25, recurrent
(GRU) MAPPO as examples/mpe/mpe_ppo.yaml runs it (use_recurrent_policy, data_chunk_length 2 and ppo_epoch from the
reference defaults), device-resident env, fused recurrent rollout (orl_rnn_rollout_fused: policy + worlds in one
launch, critic sweep in a second; --rollout stepwise = orl_rnn_act_step + orl_mpe_step + orl_buffer_insert per step as
a hipGraph) + GAE + recurrent PPO update.

    python benchmarks/cfg4_mpe_bench.py [--steps 10 --warmup 2 --envs 2048]

One JSON line: env-steps/s (N*T per iteration / wall, the reference's FPS definition - vec_info/simple_vec_info.py:30)
and the split rollout / update in ms.  Not the bench.py headline (that is config 2); an extra measurement."""
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
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--rollout", default="auto", choices=["auto", "fused", "stepwise"])
    ap.add_argument("--hidden_size", type=int, default=64, help="!= 64: the recurrent GENERAL towers (modules/generic_net.py)")
    ap.add_argument("--layer_N", type=int, default=1)
    a = ap.parse_args()
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    dev, N, T = "cuda:0", a.envs, 25
    cfg = default_cfg(["--seed", "0", "--lr", "7e-4", "--critic_lr", "7e-4", "--episode_length", str(T),
                       "--use_recurrent_policy", "true", "--use_valuenorm", "true", "--use_adv_normalize", "true",
                       "--amd_perm_mode", "device", "--amd_rollout_mode", a.rollout, "--log_interval", "1000000",
                       "--hidden_size", str(a.hidden_size), "--layer_N", str(a.layer_N)])
    env = make("simple_spread", env_num=N, device=dev)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * (a.steps + a.warmup)

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=3, device=dev)
    buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": dev}, trainer, buf,
                         _Agent())
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
    print(json.dumps({"bench": "cfg4_mpe_recurrent_mappo", "envs": N, "agents": 3, "episode_length": T,
                      "ppo_epoch": cfg.ppo_epoch, "data_chunk_length": cfg.data_chunk_length,
                      "hidden_size": a.hidden_size, "layer_N": a.layer_N,
                      "towers": "general" if getattr(net.module, "generic", False) else "fused default",
                      "rollout": "fused" if drv.fused else "stepwise (hipGraph)",
                      # 0 unless a critic workgroup's bounded wait for its policy workgroup ever timed out
                      "chase_error": int(drv._chase_flags[-1]) if getattr(drv, "_chase_flags", None) is not None else None,
                      "env_steps_per_s": N * T * a.steps / dt, "agent_steps_per_s": 3 * N * T * a.steps / dt,
                      "ms_per_iteration": dt / a.steps * 1e3, "ms_rollout": t_roll / a.steps * 1e3,
                      "ms_update": t_upd / a.steps * 1e3, "episode_return_mean": st["episode_return_mean"]}))


if __name__ == "__main__":
    main()

"""The stepwise path behind a HOST VecEnv (numpy observations in, numpy actions out - the boundary a gym /
PettingZoo / Isaac wrapper presents, SURVEY.md section 8b level 2) at the config-2 shape: 4096 envs x 128 steps,
obs 4, Discrete(2).  The env itself does no work (it returns pre-generated arrays), so the rollout time IS the boundary:
one orl_act_step + one orl_buffer_insert per step plus the host<->device copies of observations, rewards, dones and
actions over PCIe.  One JSON line; DESIGN.md quotes it as the PCIe-inclusive rate next to the device-resident headline.

    python benchmarks/host_env_bench.py [--steps 5 --warmup 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class NullHostEnv:
    """Duck-typed host VecEnv (examples/isaac/isaac2openrl.py:28-88 is the reference's precedent)."""
    env_name, use_monitor = "null-host-env", False

    def __init__(self, n, d=4, n_act=2, seed=0):
        from openrl_amd import spaces

        self.n, rs = n, np.random.RandomState(seed)
        self.observation_space = spaces.Box(-np.inf, np.inf, (d,), np.float32)
        self.action_space = spaces.Discrete(n_act)
        self._obs = [rs.randn(n, 1, d).astype(np.float32) for _ in range(8)]
        self._rew = rs.rand(n, 1, 1).astype(np.float32)
        self._done = [(rs.rand(n, 1) < 0.01) for _ in range(8)]
        self._t = 0

    parallel_env_num = property(lambda s: s.n)
    agent_num = property(lambda s: 1)

    def reset(self, seed=None, options=None):
        return self._obs[0], {}

    def step(self, actions, extra_data=None):
        self._t += 1
        return self._obs[self._t % 8], self._rew, self._done[self._t % 8], [{} for _ in range(0)]

    def batch_rewards(self, buffer):
        return {}

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.modules.common import PPONet

    dev, N, T = "cuda:0", 4096, 128
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "10", "--amd_perm_mode", "device", "--log_interval", "1000000"])
    env = NullHostEnv(N)
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
        t_roll += tb - ta
        t_upd += time.perf_counter() - tb
    dt = time.perf_counter() - t0
    per_step_bytes = N * (4 * 4 + 4 + 1 + 4)  # obs in, reward in, done in, action out
    print(json.dumps({"bench": "host_env_boundary_cfg2_shape", "envs": N, "rollout_len": T, "env_steps_per_s": N * T * a.steps / dt,
                      "ms_per_iteration": dt / a.steps * 1e3, "ms_rollout": t_roll / a.steps * 1e3,
                      "ms_update": t_upd / a.steps * 1e3, "us_per_rollout_step": t_roll / a.steps / T * 1e6,
                      "pcie_bytes_per_step": per_step_bytes}))


if __name__ == "__main__":
    main()

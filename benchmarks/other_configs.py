"""The other single-GPU BASELINE.json configs, a few iterations each, as dicts - attached to bench.py's JSON line as
``other_configs`` (outside its timed region) so the driver observes them too; also runnable by hand:

    python benchmarks/other_configs.py [--steps 3 --warmup 2]

configs[2] at its SHAPE on the synthetic env (1024 envs x 200, obs 17, Box(6); MuJoCo is not available offline),
configs[3] end to end (device MPE simple_spread, 2048 envs x 3 agents x 25, GRU, chunks of 2) and configs[4]'s env end
to end (device tic-tac-toe, 4096 envs x 200, Discrete(9) + legal-move masks, random opponent).  Reference defaults
otherwise (ppo_epoch 10, num_mini_batch 1, hidden 64).  ``dominant_kernel_ms`` = HIP-event time of one forward +
backward launch of the update (tower pair / recurrent row pair), averaged over the timed iterations."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = [
    # the metric's NAME says CartPole: configs[1] on the device CartPole-v1 PHYSICS (csrc/orl_act.hip, gymnasium's dynamics
    # and termination, in-kernel auto-reset), default tower, fused rollout - next to the headline's synthetic fixed-step env
    dict(name="configs[1] on the device CartPole-v1 physics: PPO, 4096 envs x 128, obs 4, Discrete(2), fused rollout",
         env="CartPole-v1", envs=4096, T=128, agents=1, env_kw={}, argv=[]),
    dict(name="configs[2] shape: PPO, 1024 envs x 200, obs 17, Box(6), synthetic fixed-step env", env="SyntheticFixedStep-v0",
         envs=1024, T=200, agents=1, env_kw=dict(obs_dim=17, episode_limit=200, box=6), argv=[]),
    dict(name="configs[3]: MPE simple_spread MAPPO, 2048 envs x 3 agents x 25, GRU, device env", env="simple_spread",
         envs=2048, T=25, agents=3, env_kw={},
         argv=["--lr", "7e-4", "--critic_lr", "7e-4", "--use_recurrent_policy", "true", "--use_adv_normalize", "true"]),
    dict(name="configs[4] env: tic-tac-toe vs random opponent, 4096 envs x 200, Discrete(9) + masks, device env",
         env="tictactoe_v3", envs=4096, T=200, agents=1, env_kw=dict(opponent="random"), argv=[]),
    # not a BASELINE config: the general tower path at configs[1]'s shape - the cross-layer fused
    # kernels of csrc/orl_gen_tower.h; dominant_kernel_ms = the policy tower's backward launch
    dict(name="non-default tower: configs[1]'s shape with hidden_size 128 (cross-layer fused general towers)",
         env="SyntheticFixedStep-v0", envs=4096, T=128, agents=1, env_kw=dict(obs_dim=4, episode_limit=200),
         argv=["--hidden_size", "128"]),
]


def measure(c, steps=3, warmup=2, dev="cuda:0"):
    import torch

    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    N, T, A = c["envs"], c["T"], c["agents"]
    cfg = default_cfg(["--seed", "0", "--episode_length", str(T), "--amd_perm_mode", "device", "--log_interval", "1000000"]
                      + c["argv"])
    kw = dict(c["env_kw"])
    if "box" in kw:
        kw["action_space"] = spaces.Box(-1.0, 1.0, (kw.pop("box"),))
    env = make(c["env"], env_num=N, device=dev, **kw)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
    cfg.num_env_steps = N * T * (steps + warmup + 1)

    class _Agent:
        num_time_steps = 0

    trainer = PPOAlgorithm(cfg, net.module, agent_num=A, device=dev)
    buf = NormalReplayBuffer(cfg, A, env.observation_space, env.action_space, device=dev)
    drv = OnPolicyDriver({"cfg": cfg, "num_agents": A, "run_dir": None, "envs": env, "device": dev}, trainer, buf, _Agent())
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
    out = {"workload": c["name"], "ms_per_iteration": round(1e3 * dt / steps, 4),
           "env_steps_per_s": round(N * T * steps / dt, 1), "dominant_kernel_ms": round(k_ms, 4),
           "rollout": "fused" if drv.fused else "stepwise (hipGraph)", "iterations": steps}
    try:
        out.update(pricing(cfg, env, N * T * A, k_ms, bool(getattr(trainer, "generic", False))))
    except Exception as e:  # noqa: BLE001 - the pricing must not take the measurement with it
        out["pricing_error"] = "%s: %s" % (type(e).__name__, e)
    if getattr(trainer, "recurrent", False) and not getattr(trainer, "generic", False):
        # the recurrent update launches row kernel + weight-gradient kernel + 2 reduces per epoch inside ONE C call: the events
        # bracket that group (= dominant_kernel_ms); the split between its kernels is in the rocprofv3 CSV of
        # benchmarks/cfg4_mpe_bench.py (profiles/r05_cfg4_mpe_kernel_stats.csv)
        out["ms_per_epoch_fwd_bwd_launch_group"] = round(k_ms, 4)
        out["launch_group"] = "rnn row kernel (both towers) + rnn_wgrad_kernel + rnn_reduce4_kernel, one orl_rnn_ppo_fwd_bwd call"
    return out


def pricing(cfg, env, rows, k_ms, generic):
    """Algorithmic flops of ONE forward + backward launch (group) of the update over the whole batch and its fraction of the
    fp32 MFMA peak - the same pricing as bench.py's `roofline`: per row and for both towers,
    3 x (forward flops) = forward + input gradients + weight gradients, forward = 2 x (fc1 + fc2 [+ the GRU's six 64 x 64
    products] + head) multiply-adds."""
    from openrl_amd import spaces

    H = int(cfg.hidden_size)
    osp, asp = env.observation_space, env.action_space
    if isinstance(osp, spaces.Dict):
        Dp, Dc = int(osp["policy"].shape[-1]), int(osp["critic"].shape[-1])
    else:
        Dp = Dc = int(osp.shape[-1])
    n_out = int(asp.n) if isinstance(asp, spaces.Discrete) else int(asp.shape[-1])
    fwd = 2 * ((Dp + Dc) * H + 2 * H * H + H * (n_out + 1))
    if bool(cfg.use_recurrent_policy):
        fwd += 2 * (2 * 6 * H * H)  # W_ih and W_hh, three gates each, both towers
    if generic:  # the events bracket the policy tower's backward launch only on the general path: not priced per launch
        return {"flops_fwd_per_row_both_towers": fwd, "flops_per_launch": None, "frac": None,
                "frac_note": "general tower path: dominant_kernel_ms is one tower's backward launch, not the whole forward + backward"}
    fl = 3 * fwd * rows
    return {"flops_fwd_per_row_both_towers": fwd, "flops_per_launch": fl,
            "achieved_tflops": round(fl / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else None,
            "frac": round(fl / (k_ms * 1e-3) / 157.3e12, 4) if k_ms > 0 else None,
            "frac_label": "algorithmic fp32 flops of the launch (group) / the fp32 MFMA peak 157.3 TFLOP/s"}


def run_all(steps=3, warmup=2, dev="cuda:0"):
    out = []
    for c in CONFIGS:
        try:
            out.append(measure(c, steps, warmup, dev))
        except Exception as e:  # one config failing must not take the headline line with it
            out.append({"workload": c["name"], "error": "%s: %s" % (type(e).__name__, e)})
    return out


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    for r in run_all(a.steps, a.warmup):
        print(json.dumps(r), flush=True)

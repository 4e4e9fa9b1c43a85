"""Per-phase cycle breakdown of the fused PPO tower kernel (needs the timing build:
``python -m openrl_amd.csrc.build --prof``; rebuild with ``--force`` afterwards).

    python tools/tower_phase_prof.py [--waves 8]

Wave 0 of workgroup 0 stamps the shader clock after each phase of its tile loop (csrc/orl_ppo_tower.h, ORL_T);
this prints the average cycles per tile and phase at BASELINE config 2's shape (4096 envs x 128 steps)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["dma wait/issue", "fc1+relu+LN1+store+affine", "fc2 (64 MFMA)", "LN2+store+affine+head", "loss",
          "dhead store + S3/db3 sums", "dn2 + LN2 bwd + dz2 store", "wgrad (64 MFMA) + db2", "dgrad (64 MFMA)",
          "LN1 bwd + relu + dz1 store", "dW1/db1"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--obs", type=int, default=4, help="observation width (4 = config 2; 17 / 18 = configs 3 / 5)")
    ap.add_argument("--act", type=int, default=2, help="Discrete(n)")
    ap.add_argument("--box", type=int, default=0, help="Box(k) Gaussian policy instead of Discrete(--act)")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--T", type=int, default=128)
    a = ap.parse_args()
    os.environ["ORL_KEEP_BUILD"] = "1"  # load the --prof build as it is
    from openrl_amd import _native as nat
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.ppo_module import PPOModule

    dev, N, T = "cuda:0", a.envs, a.T
    cfg = default_cfg(["--episode_length", str(T), "--ppo_epoch", "10", "--amd_perm_mode", "device"])
    cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = N, 1, cfg.hidden_size
    obs_space = spaces.Box(-np.inf, np.inf, (a.obs,))
    act_space = spaces.Box(-1.0, 1.0, (a.box,)) if a.box else spaces.Discrete(a.act)
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=dev, rank=0, world_size=1)
    buf = ReplayData(cfg, 1, obs_space, act_space, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    buf.policy_obs.copy_(torch.randn(T + 1, N, 1, a.obs, device=dev, generator=g))
    buf.rewards.copy_(torch.rand(T, N, 1, 1, device=dev, generator=g))
    buf.value_preds.copy_(0.3 * torch.randn(T + 1, N, 1, 1, device=dev, generator=g))
    if a.box:
        buf.actions.copy_(torch.randn(T, N, 1, a.box, device=dev, generator=g))
        buf.action_log_probs.fill_(-1.0)
    else:
        buf.actions.copy_(torch.randint(0, a.act, (T, N, 1, 1), device=dev, generator=g).float())
        buf.action_log_probs.fill_(float(np.log(1.0 / a.act)))
    algo = PPOAlgorithm(cfg, module, agent_num=1, device=dev)
    lib = nat.load()
    if not hasattr(lib, "orl_debug_prof"):
        raise SystemExit("liborl_hip.so is not the timing build: python -m openrl_amd.csrc.build --prof")
    out = (C.c_ulonglong * 24)()
    for it in range(3):
        buf.compute_returns(torch.zeros(N, 1, 1, device=dev), module.get_critic_value_normalizer())
        algo.train(buf)
        lib.orl_debug_prof(out)  # resets; keep the last iteration
    launches = out[12]
    tiles = out[13]  # counted by the probe wave itself (the CU split between the towers varies with the shape)
    tot = sum(out[k] for k in range(11))
    print("waves/workgroup %d: %d launches, %d tiles by the probe wave, %.0f cycles per tile" % (a.waves, launches, tiles,
                                                                                          tot / tiles))
    for k, name in enumerate(PHASES):
        print("  %-30s %8.0f  %5.1f %%" % (name, out[k] / tiles, 100.0 * out[k] / tot))
    # per launch of the probe wave: kernel entry -> first tile (staging, first record DMA) and last tile -> end (accumulator
    # reduction over the waves, partial row store); 2.4 cycles per ns at the 2.4 GHz shader clock
    print("  per launch: prologue %.0f cycles, epilogue %.0f cycles, tiles %.0f cycles (%.1f tiles)" %
          (out[14] / launches, out[15] / launches, tot / launches, tiles / launches))
    print("  epilogue, cumulative: last DMA drained %.0f, all waves arrived %.0f, accumulator images written %.0f, partial row "
          "stored %.0f" % (out[16] / launches, out[17] / launches, out[18] / launches, out[15] / launches))


if __name__ == "__main__":
    main()

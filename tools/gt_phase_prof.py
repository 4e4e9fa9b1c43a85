"""Phase timing of the fused general tower backward kernel (csrc/orl_gen_tower.h; timing build:
python -m openrl_amd.csrc.build --prof): cycles per 128-row pass of wave 0 of workgroup 0, by phase."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PHASES = ["obs + fc1 + act + LN0", "fwd HxH layers (split, GEMM, LN)", "dhead, head exchange + G3, head dgrad",
          "LN + act backward", "dz exchange (3 barriers, A reads)", "xhat store + barrier", "G_k MFMAs",
          "dgrad (split + GEMM)", "obs exchange + G0", "loop top"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden_size", type=int, default=128)
    ap.add_argument("--layer_N", type=int, default=1)
    ap.add_argument("--rows", type=int, default=524288)
    ap.add_argument("--train", action="store_true", help="orl_gt_train (losses in the kernel) instead of orl_gt_bwd")
    a = ap.parse_args()
    from openrl_amd import _native as nat, spaces
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.generic_net import GenNet

    dev = "cuda:0"
    cfg = default_cfg(["--hidden_size", str(a.hidden_size), "--layer_N", str(a.layer_N)])
    net = GenNet("policy", cfg, 4, spaces.Discrete(2), dev)
    net.host_init(cfg)
    ft = net.gt(("act",))
    B = a.rows
    x = torch.randn(B, 16, device=dev)
    idx = torch.randperm(B, device=dev)
    dh = torch.randn(B, 2, device=dev) / B
    lib = nat.load()
    if not hasattr(lib, "orl_gt_debug_prof"):
        raise SystemExit("liborl_hip.so is not the timing build: python -m openrl_amd.csrc.build --prof")
    out = (C.c_ulonglong * 16)()
    ft.prep()
    if a.train:  # the one-launch update: heads + losses inside the backward kernel (records of configs[1]'s layout)
        from openrl_amd import ops_gen
        from openrl_amd.configs.config import default_cfg as _dc

        rec = torch.randn(B, 16, device=dev)
        rec[:, 8] = torch.randint(0, 2, (B,), device=dev).float()  # action
        rec[:, 9] = -0.69                                           # old log-prob
        rec[:, 13] = 1.0                                            # active mask
        den = torch.tensor([float(B), float(B)], device=dev)
        hp = nat.PPOHParams()
        hp.clip_param, hp.value_loss_coef, hp.entropy_coef, hp.huber_delta = 0.2, 0.5, 0.01, 10.0
        hp.use_clipped_value_loss, hp.use_huber_loss = 1, 1
        head = ops_gen.head_desc(ops_gen.HEAD_CATEGORICAL, 2)
    for it in range(3):
        if a.train:
            ft.train(rec, 0, idx, B, head, None, 4, 4, 1, 0, den, None, hp)
        else:
            ft.backward(x, 0, idx, B, dh)
        lib.orl_gt_debug_prof(out)
    passes = out[12]
    tot = sum(out[k] for k in range(10))
    print("hidden %d layer_N %d: %d passes by the probe wave, %.0f cycles (s_memtime) per pass" % (a.hidden_size, a.layer_N,
                                                                                                passes, tot / passes))
    for k, name in enumerate(PHASES):
        print("  %-42s %8.0f  %5.1f %%" % (name, out[k] / passes, 100.0 * out[k] / tot))


if __name__ == "__main__":
    main()

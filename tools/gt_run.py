"""Runs the fused general tower kernels a few times on a synthetic batch (profiling driver for tools/pmc_gt.sh)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden_size", type=int, default=128)
    ap.add_argument("--layer_N", type=int, default=1)
    ap.add_argument("--rows", type=int, default=524288)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from openrl_amd import spaces
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
    out = torch.empty(B, 2, device=dev)
    ft.prep()
    for _ in range(a.iters):
        ft.forward(x, 0, idx, B, out)
        ft.backward(x, 0, idx, B, dh)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()

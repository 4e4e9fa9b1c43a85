"""Latency of the rollout's one-launch tower (``orl_gen_mlp_fwd``) by width, depth and rows, 50 launches per hipGraph
replay: 16 rows (one workgroup) cost what 4096 rows cost - the tile is a latency chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openrl_amd import spaces, ops_gen
from openrl_amd.configs.config import default_cfg
from openrl_amd.modules import generic_net as gn
DEV = "cuda:0"
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for H in (64, 128, 256):
  for ln in (1, 2, 4):
    cfg = default_cfg(["--hidden_size", str(H), "--layer_N", str(ln)])
    mod = gn.GenericPPOModule(cfg, spaces.Box(-np.inf, np.inf, (4,)), spaces.Box(-np.inf, np.inf, (4,)), spaces.Discrete(2), share_model=False, device=DEV)
    pn = mod.policy_net
    desc = pn.mlp_desc(("act",))
    for B in (16, 4096):
        x = torch.randn(B, 4, device=DEV); out = torch.zeros(B, 2, device=DEV)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops_gen.mlp_fwd(desc, x, out)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(50): ops_gen.mlp_fwd(desc, x, out)
        us = t(g.replay, 20) / 50
        print(f"H {H} layer_N {ln} B {B}: {us:.2f} us per launch (graph of 50)", flush=True)

"""Per-phase cycle breakdown of the recurrent row kernel (timing build: python -m openrl_amd.csrc.build --prof).
Runs benchmarks/rnn_update_bench.py's workload (cfg4 shape) and prints the probe wave's cycles per backward step."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PH = ["forward sweep (per tile)", "step inputs (loads)", "trunk recompute + tapes", "GRU forward (384 MFMA)", "LN3 + head + loss",
      "dhead/obs tape, W3^T, LN3 bwd", "GRU elementwise bwd + tapes", "GRU dgrad (384 MFMA)", "LN2 bwd, W2 dgrad, LN1 bwd, tapes"]


def main():
    import torch
    from openrl_amd import _native as nat
    gemm = sys.argv[1] if len(sys.argv) > 1 else "split"  # split | fp32 | split_w4
    sys.argv = [sys.argv[0], "--iters", "1", "--warmup", "1", "--epochs", "4", "--tower-gemm", gemm]
    lib = nat.load()
    if not hasattr(lib, "orl_debug_rnn_prof"):
        raise SystemExit("not the timing build")
    import benchmarks.rnn_update_bench as b
    out = (C.c_ulonglong * 16)()
    lib.orl_debug_rnn_prof(out)
    b.main()
    lib.orl_debug_rnn_prof(out)
    launches = out[12]
    tiles = launches * 4800 // (128 * 8)          # tiles of the probe wave (policy tower: 128 workgroups x 8 waves)
    if out[13]:                                   # the streamed kernel counts its probe wave's tile iterations itself
        tiles = out[13]
    print("row kernel GEMM path:", gemm)
    if out[14]:  # the register-resident L = 2 kernel (csrc/orl_rnn_l2.h): phases per TILE (two steps)
        names = {1: "input hand-over (waits for the prefetched loads)", 2: "trunk fwd + tapes (x2)", 3: "GRU forward (x2)",
                 4: "LN3 + head + loss + W3^T + LN3' (x2)", 6: "gate deltas + 4 tape vectors (x2)",
                 7: "hidden-state dgrad (step 1 only)", 8: "W_ih^T dgrad, LN2', W2^T, LN1', relu', tapes (x2)"}
        tot = sum(out[k] for k in names)
        print("L = 2 register-resident kernel: %d launches, %d tiles by the probe wave; %.0f cycles per tile" % (launches, tiles, tot / max(tiles, 1)))
        for k, n in names.items():
            print("  %-56s %8.0f cycles per tile  %5.1f %%" % (n, out[k] / max(tiles, 1), 100.0 * out[k] / tot))
        return
    steps = tiles * 2
    tot = sum(out[k] for k in range(9))
    print("%d launches, ~%d tiles / %d backward steps by the probe wave; %.0f cycles per tile" % (launches, tiles, steps, tot / max(tiles, 1)))
    for k, n in enumerate(PH):
        per = out[k] / max(tiles if k == 0 else steps, 1)
        print("  %-36s %8.0f cycles   %5.1f %%" % (n, per, 100.0 * out[k] / tot))


if __name__ == "__main__":
    main()

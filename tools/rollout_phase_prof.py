"""Per-phase cycle breakdown of the fused rollout kernel (timing build: python -m openrl_amd.csrc.build --prof;
rebuild with --force afterwards).  Waves 0 (policy leader) and 1 (env wave of the synthetic env) of workgroup 0."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["fc1 + gather store", "barrier 1", "LN1 + fc2 + gather store", "barrier 2", "LN2 + affine",
          "head + sample + stores (wave 0)", "env step / value head", "barrier 3"]


def main():
    import torch

    from openrl_amd import _native as nat
    import bench

    lib = nat.load()
    if not hasattr(lib, "orl_debug_rollout_prof"):
        raise SystemExit("liborl_hip.so is not the timing build: python -m openrl_amd.csrc.build --prof")
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-other-configs", "--steps", "5", "--warmup", "2"]
    out = (C.c_ulonglong * 16)()
    lib.orl_debug_rollout_prof.argtypes = [C.c_void_p]
    bench.main()
    torch.cuda.synchronize()
    lib.orl_debug_rollout_prof(out)
    v = list(out)
    steps = 7 * 128
    for w in (0, 1):
        tot = sum(v[8 * w:8 * w + 8])
        print("wave %d: %.0f cycles/step" % (w, tot / steps))
        for k, name in enumerate(PHASES):
            print("   %-36s %7.0f  %5.1f %%" % (name, v[8 * w + k] / steps, 100.0 * v[8 * w + k] / max(tot, 1)))


if __name__ == "__main__":
    main()

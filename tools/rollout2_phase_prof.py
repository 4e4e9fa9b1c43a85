"""Per-phase cycle breakdown of the round-6 chain rollout kernel (csrc/orl_rollout2.h), wave 0 of workgroup 0.  Timing build:
python -m openrl_amd.csrc.build --prof (or variants/prof.so copied over liborl_hip.so with ORL_KEEP_BUILD=1).
    python tools/rollout2_phase_prof.py [--env cartpole] [--envs N] | --shape cfg3 | --shape cfg5"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["wait for observation t (+ operand)", "fc1 + relu", "LayerNorm 1", "fc2 (16 MFMA)", "partials + hand-over",
          "batch: flags + partials + noise + env record", "combine: LayerNorm 2 + head", "sample",
          "env post-step + stage + publish"]


def main():
    import torch

    from openrl_amd import _native as nat
    import bench

    lib = nat.load()
    if not hasattr(lib, "orl_debug_rollout_prof"):
        raise SystemExit("liborl_hip.so is not the timing build: python -m openrl_amd.csrc.build --prof")
    extra = sys.argv[1:]
    out = (C.c_ulonglong * 16)()
    lib.orl_debug_rollout_prof.argtypes = [C.c_void_p]
    lib.orl_debug_rollout_prof(out)  # reset
    if extra[:1] == ["--shape"]:  # --shape cfg3 | cfg5: benchmarks/shape_sweep.py's shapes on the synthetic env (8 rollouts)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
        import shape_sweep

        name = [k for k in shape_sweep.SHAPES if extra[1] in k][0]
        shape_sweep.run(name, shape_sweep.SHAPES[name], 5, 2)
        steps = 8 * shape_sweep.SHAPES[name]["T"]
    else:
        sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-other-configs", "--steps", "5", "--warmup", "2"] + extra
        bench.main()
        steps = 7 * 128
    torch.cuda.synchronize()
    lib.orl_debug_rollout_prof(out)
    v = list(out)
    tot = sum(v[:9])
    print("wave 0: %.0f cycles/step" % (tot / steps))
    for k, name in enumerate(PHASES):
        print("   %-40s %7.0f  %5.1f %%" % (name, v[k] / steps, 100.0 * v[k] / max(tot, 1)))


if __name__ == "__main__":
    main()

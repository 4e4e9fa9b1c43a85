"""Per-phase cycle breakdown of the fused rollout kernel on the device tic-tac-toe env (timing build:
python -m openrl_amd.csrc.build --prof; run with ORL_KEEP_BUILD=1).  Usage: rollout_phase_prof_ttt.py [random|pool] [sampling]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PH = ["fc1 + gather store", "barrier 1", "LN1 + fc2 + gather store", "barrier 2", "LN2 + affine",
      "head + sample (+ opponent's move)", "env step / value head", "barrier 3"]


def main():
    import torch
    from openrl_amd import _native as nat
    import benchmarks.cfg5_ttt_bench as b

    opp = sys.argv[1] if len(sys.argv) > 1 else "random"
    samp = sys.argv[2] if len(sys.argv) > 2 else "per_reset"
    lib = nat.load()
    lib.orl_debug_rollout_prof.argtypes = [C.c_void_p]
    out = (C.c_ulonglong * 16)()
    steps, warm, T = 4, 2, 200
    sys.argv = [sys.argv[0], "--steps", str(steps), "--warmup", str(warm), "--opponent", opp, "--sampling", samp]
    lib.orl_debug_rollout_prof(out)
    b.main()
    torch.cuda.synchronize()
    lib.orl_debug_rollout_prof(out)
    v, n = list(out), (steps + warm) * T
    for w in (0, 1):
        tot = sum(v[8 * w:8 * w + 8])
        print("wave %d: %.0f cycles/step" % (w, tot / n))
        for k, nm in enumerate(PH):
            print("   %-36s %7.0f" % (nm, v[8 * w + k] / n))


if __name__ == "__main__":
    main()

import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from openrl_amd import _native as nat
from benchmarks import shape_sweep as ss
lib = nat.load()
lib.orl_debug_rollout_prof.argtypes = [C.c_void_p]
out = (C.c_ulonglong * 16)()
name = sys.argv[1]
steps, warm = 5, 2
lib.orl_debug_rollout_prof(out)  # reset
print(ss.run(name, ss.SHAPES[name], steps, warm))
torch.cuda.synchronize()
lib.orl_debug_rollout_prof(out)
v = list(out)
T = ss.SHAPES[name]["T"]
n = (steps + warm) * T
PH = ["fc1 + gather store", "barrier 1", "LN1 + fc2 + gather store", "barrier 2", "LN2 + affine", "head + sample + stores (wave 0)", "env step / value head", "barrier 3"]
for w in (0, 1):
    tot = sum(v[8*w:8*w+8]); print("wave %d: %.0f cycles/step" % (w, tot / n))
    for k, nm in enumerate(PH): print("   %-36s %7.0f" % (nm, v[8*w+k] / n))

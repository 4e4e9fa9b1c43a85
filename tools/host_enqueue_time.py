"""Host time per training iteration inside bench.py's timed loop (how far the CPU runs ahead of the GPU).

    python tools/host_enqueue_time.py

Measured on the round-6 tree: 0.41 ms of host work per iteration (median 0.39, max 0.64) against 2.03 ms of GPU work - the launch
queue deepens by ~1.6 ms per iteration, so host-side hiccups of tens of milliseconds do not reach the device."""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import bench
# monkeypatch: measure host time of _inner_loop calls inside bench.main()
from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
orig = OnPolicyDriver._inner_loop
times = []
def timed(self):
    t0 = time.perf_counter(); r = orig(self); times.append(time.perf_counter() - t0); return r
OnPolicyDriver._inner_loop = timed
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-configs", "--steps", "40", "--warmup", "3"]
bench.main()
import numpy as np
t = np.array(times[3:43]) * 1e3
print("host ms per _inner_loop: mean %.3f median %.3f max %.3f p90 %.3f" % (t.mean(), np.median(t), t.max(), np.percentile(t, 90)))

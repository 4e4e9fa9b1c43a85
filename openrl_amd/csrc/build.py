"""Build ``liborl_hip.so`` (all HIP kernels + the C ABI) for gfx950, in-tree.

``python -m openrl_amd.csrc.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without
a GPU.  The shared object is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["orl_buffer.hip", "orl_act.hip", "orl_ppo.hip", "orl_apply.hip", "orl_rnn.hip", "orl_rnn_rollout.hip", "orl_mpe.hip", "orl_ttt.hip", "orl_comm.hip", "orl_gen.hip", "orl_gen_fused.hip", "orl_gen_rollout.hip", "orl_gen_tower.hip", "orl_gen_tower128.hip"]
# every header of this directory enters the up-to-date digest (a hand-kept list missed orl_rnn_l2.h and orl_rnn_rollout_coop.h: an
# edit of either left a stale library behind unless the build was forced)
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join("..", "..", "include", "orl_hip.h")]
# orl_mpe.hip shares orl_mpe.h with the fused recurrent rollout: same flags, so both step a world with the same code
FAST_MATH = {"orl_ppo.hip", "orl_rnn.hip", "orl_rnn_rollout.hip", "orl_mpe.hip", "orl_act.hip", "orl_gen_rollout.hip"}  # NOT orl_gen.hip: torch-like IEEE arithmetic there
LIB = os.path.join(HERE, "liborl_hip.so")
STAMP = os.path.join(HERE, ".liborl_hip.stamp")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build the gfx950 extension")


def have_hipcc() -> bool:
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, prof: bool = False) -> str:
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    # build-time A/B switches for kernel experiments (e.g. ORL_BUILD_DEFS="-DORL_TOWER_ILV"); never read at run time
    extra = os.environ.get("ORL_BUILD_DEFS", "").split()
    compress = os.environ.get("ORL_NO_OFFLOAD_COMPRESS", "") == ""
    dig = _digest() + ("-prof" if prof else "") + ("-oc" if compress else "") + "".join(extra)  # a library built with other switches is not up to date
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    # Per-source flags: the update and rollout kernels (towers, recurrent, act) are built with -ffast-math -
    # reassociation and 1-ulp hardware reciprocal / rsqrt / exp / sin / cos instead of the IEEE sequences (+5 % on
    # the bench iteration, every parity test unchanged); the buffer kernels are NOT (the GAE scan is bit-exact with
    # the reference's operation order) and neither is the MPE physics (compared with a float64 reference).
    base = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    # the gfx950 code objects are stored compressed in the fat binary (10.3 -> 2.75 MB; the HIP runtime inflates them when the
    # library is loaded: no difference in load time, first-launch time or anything after - tools/r06_calls/r06_call13.sh)
    if compress:
        base.append("--offload-compress")
    base += extra
    if prof:  # phase-timing build of the tower kernels (orl_debug_prof); never the shipped configuration
        base.append("-DORL_PROF")
    def compile_one(src):
        obj = os.path.splitext(src)[0] + ".o"
        cmd = base + (["-ffast-math"] if os.path.basename(src) in FAST_MATH else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        return obj

    # the translation units are independent: compile them side by side (wall time = the slowest one, orl_ppo.hip)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    for obj in objs:
        if os.path.exists(obj):
            os.remove(obj)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--prof" in sys.argv, verbose=True, prof="--prof" in sys.argv))

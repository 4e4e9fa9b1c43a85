#!/usr/bin/env python
"""Register budget of the fused tower kernel per phase, read off the ISA (round-5 VERDICT item 2: "if a third wave per SIMD
cannot fit, commit the register / LDS budget table that shows why, per live range").

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -DORL_MARK --cuda-device-only -S \\
        openrl_amd/csrc/orl_ppo.hip -o /tmp/ppo_mark.s
    python tools/tower_live_ranges.py /tmp/ppo_mark.s [kernel-substring]

For every tower body of the kernel (the pair kernel holds two: policy, critic) a backward liveness pass over the tile loop
in cyclic text order (the loop is treated as straight-line code: its few forward branches skip at most a handful of
instructions, so the result is an upper bound that is tight to a few registers):

  * LOOP-CARRIED registers = live across the back edge (accumulators, ring state, addresses, hoisted operands);
  * per phase (tools/tower_valu_budget.py's markers): the peak number of live VGPRs and how many of them are not loop-carried
    (the phase's own working set);
  * the longest-lived values born inside a tile (defined in one phase, last used several phases later): what a smaller
    per-wave footprint would have to re-load or re-compute.

Runs on the CPU; needs no GPU."""
import re
import sys

from tower_valu_budget import PHASES, bodies, kernels

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


NO_DST = ("ds_write", "ds_add", "global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "v_cmp",
          "v_cmpx", "s_", "ds_wakeup", "global_load_lds", "buffer_load")  # (buffer_load ... lds: no VGPR destination in this kernel)
BOTH = ("v_permlane16_swap", "v_permlane32_swap", "v_swap")


def def_use(text):
    """(defs, uses) of VGPRs of one instruction line"""
    t = text.split(";")[0].strip()
    if not t:
        return set(), set()
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    if not ops:
        return set(), set()
    if op.startswith(BOTH):
        r = regs_of(ops[0]) | regs_of(ops[1])
        return r, r
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        return set(), set().union(*[regs_of(o) for o in ops[1:]])
    if op.startswith(NO_DST):
        return set(), set().union(*[regs_of(o) for o in ops])
    d = regs_of(ops[0])
    u = set().union(*[regs_of(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac", "v_writelane")) or "op_sel" in t or "dpp" in t or "sdwa" in t:
        u |= d  # the destination is also read
    return d, u


def analyse(klines, body):
    lo, end, marks = body
    pos10 = next(i for i, k in marks if k == 10)
    order = list(range(pos10 + 1, end)) + list(range(lo, pos10 + 1))
    where = {i: k for i, k in marks}
    instrs, phase_of = [], []
    cur = 0
    for i in order:
        if i in where:
            cur = where[i] + 1
            continue
        t = klines[i].strip()
        if not t or t.startswith((";", ".", "//")) or re.match(r"^\.?\w+:", t):
            continue
        instrs.append(def_use(t))
        phase_of.append(min(cur, 10))
    n = len(instrs)
    live_out = set()
    for _ in range(3):  # fixpoint over the back edge
        live = set(live_out)
        per = [None] * n
        for k in range(n - 1, -1, -1):
            d, u = instrs[k]
            per[k] = set(live)
            live = (live - d) | u
        live_out = live
    carried = live_out  # live at the loop top = across the back edge
    rows = {}
    for k in range(n):
        p = phase_of[k]
        cnt = len(per[k])
        own = len(per[k] - carried)
        r = rows.setdefault(p, [0, 0, 0])
        if cnt > r[0]:
            r[0], r[1] = cnt, own
        r[2] += 1
    # values born inside a tile: (def index -> last use index) in the cyclic order
    spans = []
    last_use = {}
    for k in range(n - 1, -1, -1):
        d, u = instrs[k]
        for r in d:
            if r in last_use and r not in carried:
                spans.append((phase_of[k], phase_of[last_use[r]], last_use[r] - k))
                del last_use[r]
        for r in u:
            last_use.setdefault(r, k)
    return carried, rows, spans, n


def main():
    path = sys.argv[1]
    pats = sys.argv[2:] or ["ppo_tower_pair_kernelILi1ELi2ELi0ELi2E"]
    lines = open(path).read().splitlines()
    for name, kl in kernels(lines):
        if not any(p in name for p in pats):
            continue
        alloc = [ln.strip() for ln in kl if ".vgpr_count" in ln or "ScratchSize" in ln or ".vgpr_spill_count" in ln]
        print("==", name)
        for ln in alloc[:4]:
            print("   ", ln)
        for b, body in enumerate(bodies(kl)):
            carried, rows, spans, n = analyse(kl, body)
            print("  body %d: %d instructions in the tile loop, %d VGPRs live across the back edge (loop-carried)" % (b, n, len(carried)))
            print("    %-24s %10s %18s %8s" % ("phase", "peak live", "of them per-tile", "instrs"))
            for p in sorted(rows):
                pk, own, cnt = rows[p]
                print("    %-24s %10d %18d %8d" % (PHASES[p], pk, own, cnt))
            peak = max(r[0] for r in rows.values())
            print("    peak over the tile: %d (allocation granule 8: %d)" % (peak, (peak + 7) // 8 * 8))
            far = {}
            for p0, p1, dist in spans:
                if p1 != p0:
                    key = (p0, p1)
                    far[key] = far.get(key, 0) + 1
            print("    values born in one phase and last read in another (registers x phases they stay live):")
            for (p0, p1), c in sorted(far.items(), key=lambda kv: -kv[1] * ((kv[0][1] - kv[0][0]) % 11))[:10]:
                print("      %3d registers  %-22s -> %-22s (%d phases)" % (c, PHASES[p0], PHASES[p1], (p1 - p0) % 11))


if __name__ == "__main__":
    main()

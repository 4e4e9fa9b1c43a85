#!/usr/bin/env python
"""Per-phase instruction budget of the fused tower kernel, read off the ISA (round-3 VERDICT item 2c).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -DORL_MARK --cuda-device-only -S \\
        openrl_amd/csrc/orl_ppo.hip -I include -o /tmp/ppo_mark.s
    python tools/tower_valu_budget.py /tmp/ppo_mark.s [kernel-substring ...]

`-DORL_MARK` turns the ORL_T(k) phase boundaries of csrc/orl_ppo_tower.h into volatile asm comments ("; ORL_PHASE k"); the
pair kernel holds two bodies (policy, critic), each with one tile loop.  For every body the instructions between marker
k-1 and marker k INSIDE the tile loop are attributed to phase k (phase 0 = loop top .. marker 0) and classified:
MFMA / VALU (other v_*) / LDS (ds_*) / VMEM (global_, buffer_, scratch_) / SALU (s_* except s_waitcnt, s_nop, s_barrier) /
WAIT (s_waitcnt).  Scratch traffic inside the loop is listed separately (= spills actually paid per tile).
The markers order the phases but are no scheduling barriers for register-only instructions: hipcc moves e.g. an operand
split into the neighbouring phase, so single rows are +-40 instructions - the TOTAL per tile is exact.
Runs on the CPU; needs no GPU."""
import re
import sys
from collections import OrderedDict

PHASES = ["0 dma wait+issue", "1 fc1 relu LN1 store", "2 fc2", "3 LN2 store head", "4 loss", "5 dhead S3 db3",
          "6 dn2 LN2' store", "7 wgrad db2", "8 dgrad", "9 LN1' relu' store", "10 dW1 db1"]


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "VMEM"
    if op.startswith("scratch_"):
        return "SCRATCH"
    if op == "s_waitcnt":
        return "WAIT"
    if op in ("s_nop", "s_barrier", "s_endpgm", "s_branch") or op.startswith("s_cbranch"):
        return "CTRL"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def kernels(lines):
    name, start = None, 0
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            if name:
                yield name, lines[start:i]
            name, start = m.group(1), i
    if name:
        yield name, lines[start:]


def bodies(klines):
    """one tile loop per tower body: the outermost loop (Depth=1 header) that contains a full set of markers 0..10"""
    heads = [(i, m.group(1)) for i, ln in enumerate(klines)
             for m in [re.match(r"^\.(LBB\w+):.*=>This Loop Header: Depth=1", ln)] if m]
    out = []
    for hi, label in heads:
        member = [i for i, ln in enumerate(klines) if re.match(r"^\.LBB\w+:", ln) and
                  ("Header=%s " % label[1:] in ln or "Header=%s\n" % label[1:] in ln + "\n" or i == hi)]
        lo, hi2 = min(member), max(member)
        # the loop's last block runs to the next label that is not part of it
        end = hi2 + 1
        while end < len(klines) and not re.match(r"^\.LBB\w+:", klines[end]) and not klines[end].startswith(".Lfunc_end"):
            end += 1
        marks = [(i, int(m.group(1))) for i in range(lo, end) for m in [re.search(r"; ORL_PHASE (\d+)", klines[i])] if m]
        if sorted(k for _, k in marks) == list(range(11)):
            out.append((lo, end, marks))
    return out


def budget(klines, body):
    """instructions of the loop in CYCLIC text order starting after marker 10 (hipcc rotates the loop: the text order is
    5 .. 10, header, 0 .. 4); phase k = everything between marker k-1 and marker k"""
    lo, end, marks = body
    pos10 = next(i for i, k in marks if k == 10)
    order = list(range(pos10 + 1, end)) + list(range(lo, pos10 + 1))
    where = {i: k for i, k in marks}
    table = OrderedDict((k, {}) for k in range(11))
    cur = 0
    for i in order:
        if i in where:
            cur = where[i] + 1
            continue
        t = klines[i].strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or (t.startswith(".") and ":" in t.split()[0]):
            continue
        if re.match(r"^\.?\w+:", t):
            continue
        c = classify(t.split()[0])
        if cur <= 10:
            table[cur][c] = table[cur].get(c, 0) + 1
    return table


def main():
    path = sys.argv[1]
    pats = sys.argv[2:] or ["ppo_tower_pair_kernelILi1ELi2ELi0ELi2E"]
    lines = open(path).read().splitlines()
    for name, kl in kernels(lines):
        if not any(p in name for p in pats):
            continue
        vg = [ln for ln in kl if ".vgpr_count" in ln or "NumVgprs" in ln or "ScratchSize" in ln or "vgpr_spill" in ln]
        print("==", name)
        for ln in vg[:6]:
            print("   ", ln.strip())
        for b, body in enumerate(bodies(kl)):
            tab = budget(kl, body)
            cols = ["VALU", "MFMA", "LDS", "SALU", "WAIT", "VMEM", "SCRATCH", "CTRL"]
            # (hipcc lays the `else` branch of the pair kernel out first: judge the tower by its loss / head rows - the value
            # head's loss is ~55 VALU, a categorical head's 100 - 280)
            print("  body %d (in ISA order)" % b)
            print("    %-24s" % "phase" + "".join("%8s" % c for c in cols))
            tot = dict.fromkeys(cols, 0)
            for k, cnt in tab.items():
                print("    %-24s" % PHASES[k] + "".join("%8d" % cnt.get(c, 0) for c in cols))
                for c in cols:
                    tot[c] += cnt.get(c, 0)
            print("    %-24s" % "TOTAL per tile" + "".join("%8d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Opcode histogram per phase of the tower kernel's tile loop (same marked ISA as tools/tower_valu_budget.py), and the loop's
text per phase written to <out_dir>/body<b>_phase<k>.s for reading.  CPU only.

    python tools/tower_isa_hist.py build_tools/ppo_mark.s ILi1ELi2ELi0ELi2E [out_dir]
"""
import os
import re
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tower_valu_budget import PHASES, bodies, classify, kernels  # noqa: E402


def main():
    path, pat = sys.argv[1], sys.argv[2]
    out_dir = sys.argv[3] if len(sys.argv) > 3 else None
    lines = open(path).read().splitlines()
    for name, kl in kernels(lines):
        if pat not in name:
            continue
        print("==", name)
        for b, (lo, end, marks) in enumerate(bodies(kl)):
            pos10 = next(i for i, k in marks if k == 10)
            order = list(range(pos10 + 1, end)) + list(range(lo, pos10 + 1))
            where = {i: k for i, k in marks}
            cur, hist, text = 0, {k: Counter() for k in range(11)}, {k: [] for k in range(11)}
            for i in order:
                if i in where:
                    cur = where[i] + 1
                    continue
                t = kl[i].strip()
                if cur > 10:
                    continue
                text[cur].append(kl[i])
                if not t or t.startswith((";", ".", "//")) or t.endswith(":") or re.match(r"^\.?\w+:", t):
                    continue
                op = t.split()[0]
                if classify(op) in ("VALU",):
                    hist[cur][re.sub(r"_e32|_e64|_dpp|_sdwa", "", op)] += 1
            total = Counter()
            print("  body %d" % b)
            for k in range(11):
                total.update(hist[k])
                print("    %-22s %4d  " % (PHASES[k], sum(hist[k].values())) +
                      " ".join("%s:%d" % (o.replace("v_", ""), c) for o, c in hist[k].most_common(14)))
                if out_dir:
                    os.makedirs(out_dir, exist_ok=True)
                    open(os.path.join(out_dir, "body%d_phase%d.s" % (b, k)), "w").write("\n".join(text[k]) + "\n")
            print("    TOTAL %d  " % sum(total.values()) + " ".join("%s:%d" % (o.replace("v_", ""), c) for o, c in total.most_common(30)))


if __name__ == "__main__":
    main()

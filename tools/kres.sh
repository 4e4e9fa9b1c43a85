#!/bin/bash
# kernel resource usage of one translation unit (registers, scratch, LDS, occupancy) from hipcc's remarks; runs on the CPU.
#   tools/kres.sh orl_ppo.hip [extra -D flags]   -> one line per kernel: name vgpr agpr sgpr scratch occ lds
src=$1; shift
cd "$(dirname "$0")/../openrl_amd/csrc"
fm=""
case "$src" in orl_ppo.hip|orl_rnn.hip|orl_rnn_rollout.hip|orl_mpe.hip|orl_act.hip|orl_gen_rollout.hip) fm="-ffast-math";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $fm "$@" --cuda-device-only -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    s = m.group(1).strip()
    if s.startswith("Function Name:"):
        cur = {"name": s.split(":", 1)[1].strip()}
        rows.append(cur)
    for k, key in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("TotalSGPRs:", "sgpr"), ("ScratchSize [bytes/lane]:", "scratch"),
                   ("LDS Size [bytes/block]:", "lds"), ("Occupancy [waves/SIMD]:", "occ"), ("VGPRs Spill:", "vspill")):
        if s.startswith(k):
            cur[key] = s.split(":", 1)[1].strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    print(n[:100], "vgpr", r.get("vgpr"), "agpr", r.get("agpr"), "sgpr", r.get("sgpr"), "scratch", r.get("scratch"), "vspill", r.get("vspill"), "occ", r.get("occ"), "lds", r.get("lds"))
'

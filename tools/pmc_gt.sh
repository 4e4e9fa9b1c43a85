#!/usr/bin/env bash
# PMC passes on the fused general tower kernels: bash tools/pmc_gt.sh [args of tools/gt_run.py]   (through gpurun)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcgt_$i -- python tools/gt_run.py "$@" > gpurun_out/pmcgt_$i.log 2>&1
  f=$(find /tmp/pmcgt_$i -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "no counter file for: $C"; tail -3 gpurun_out/pmcgt_$i.log; continue; fi
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    for tag in ("gt_bwd_kernel","gt_fwd_kernel"):
        if tag in k:
            acc[(tag,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k[0], k[1], round(sum(v)/len(v),1), len(v))
PY
done

#!/usr/bin/env bash
# PMC passes on the fused tower kernel (bench.py workload): bash tools/pmc_tower.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmct_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/pmct_$i.log 2>&1
  f=$(find gpurun_out/pmct_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r["Kernel_Name"]
        if "ppo_tower_pair_kernel" in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, round(sum(v)/len(v),1), len(v))
except Exception as e: print("ERR", e)
PY
  rm -rf gpurun_out/pmct_$i
done

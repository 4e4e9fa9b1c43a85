# usage: bash tools/pmc_rnn_row.sh [split|fp32|split_w4]   (the row kernel's GEMM path, benchmarks/rnn_update_bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GEMM=${1:-split}
echo "row kernel GEMM path: $GEMM"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmcrow_$i -- python benchmarks/rnn_update_bench.py --iters 1 --warmup 0 --epochs 2 --tower-gemm $GEMM > gpurun_out/pmcrow_$i.log 2>&1
  f=$(find gpurun_out/pmcrow_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r["Kernel_Name"]
        if "rnn_row" in k or "rnn_wgrad" in k:
            acc[("row" if "rnn_row" in k else "wgrad", r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, round(sum(v)/len(v),1), len(v))
except Exception as e: print("ERR", e)
PY
  rm -rf gpurun_out/pmcrow_$i
done

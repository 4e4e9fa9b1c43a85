#!/usr/bin/env bash
# LDS counters of the recurrent update's kernels after the tape's rotation change (compare with profiles/r05_pmc_rnn.txt): two --pmc passes
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c26
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python benchmarks/rnn_update_bench.py --iters 1 --warmup 0 --epochs 2 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' | tee -a $OUT/pmc_rnn_lds.txt
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "rnn_row" in k or "rnn_wgrad" in k:
        acc[("row" if "rnn_row" in k else "wgrad", r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, round(sum(v)/len(v),1), len(v))
PY
  rm -rf $OUT/p$i
done

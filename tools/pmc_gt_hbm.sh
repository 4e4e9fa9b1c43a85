#!/usr/bin/env bash
# HBM traffic of the fused general tower kernels: separate --pmc FETCH_SIZE / WRITE_SIZE passes (KiB per launch, means).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcgth_$C -- python tools/gt_run.py "$@" > gpurun_out/pmcgth_$C.log 2>&1
  f=$(find /tmp/pmcgth_$C -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "no counter file for $C"; tail -3 gpurun_out/pmcgth_$C.log; continue; fi
  python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    for tag in ("gt_bwd_kernel","gt_fwd_kernel","gt_prep_kernel","gt_finalize_kernel","gen_colsum_kernel"):
        if tag in k:
            acc[(tag,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k[0], k[1], "KiB per launch (mean)", round(sum(v)/len(v),1), len(v))
PY
done

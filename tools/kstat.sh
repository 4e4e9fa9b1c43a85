#!/bin/bash
# usage: kstat.sh <tag> <cmd...>  -> prints top kernels
tag=$1; shift
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$tag -- "$@" > gpurun_out/st_$tag.log 2>&1
f=$(find gpurun_out/st_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:9]:
    print(r[0][:50].ljust(50), r[1], r[3][:9], r[5], r[6])
PY

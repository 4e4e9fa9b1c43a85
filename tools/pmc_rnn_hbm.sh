#!/usr/bin/env bash
# HBM traffic of the recurrent update's two kernels at the cfg4 shape: FETCH_SIZE / WRITE_SIZE, one --pmc pass each (through gpurun)
#   bash tools/pmc_rnn_hbm.sh [fp32|fp32_recompute|split]
# Corrections as tools/summarize_pmc.py (profiles/r02_pmc_calibration.json): WRITE_SIZE exact; FETCH_SIZE tallies a coalesced
# stream (the wgrad kernel's 16-B global_load_lds DMA of the tape, the row kernel's streamed loads) at 1/2.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GEMM=${1:-fp32}
echo "row kernel: $GEMM (benchmarks/rnn_update_bench.py, 2 epochs; KiB per launch as counted, then corrected MB per launch / per epoch)"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmcrh_$C -- python benchmarks/rnn_update_bench.py --iters 1 --warmup 0 --epochs 2 --tower-gemm $GEMM > gpurun_out/pmcrh_$C.log 2>&1
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmcrh_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if r["Counter_Name"] == C and ("rnn_row" in k or "rnn_wgrad" in k):
                acc[("row" if "rnn_row" in k else "wgrad", C)].append(float(r["Counter_Value"]))
tot = 0.0
for kern in ("row", "wgrad"):
    f = sum(acc[(kern, "FETCH_SIZE")]) / max(len(acc[(kern, "FETCH_SIZE")]), 1)
    w = sum(acc[(kern, "WRITE_SIZE")]) / max(len(acc[(kern, "WRITE_SIZE")]), 1)
    mb = (2.0 * f + w) * 1024 / 1e6
    tot += mb
    print("%-6s FETCH_SIZE %.0f KiB  WRITE_SIZE %.0f KiB per launch (%d launches) -> 2 x FETCH + WRITE = %.1f MB per launch" % (kern, f, w, len(acc[(kern, "FETCH_SIZE")]), mb))
print("both kernels: %.1f MB per epoch (tape written once by the row kernel, read once by the wgrad kernel); SURVEY 8d's S_upd for this shape is 51 MB" % tot)
PY
rm -rf gpurun_out/pmcrh_FETCH_SIZE gpurun_out/pmcrh_WRITE_SIZE

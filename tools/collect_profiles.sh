#!/usr/bin/env bash
# Collect the round's measurements on a MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Writes everything under gpurun_out/<tag>_*; copy what is to be judged into profiles/ (tracked).
# Counter passes are separate runs with --pmc only (no sys/hip/hsa traces), as the pool requires.
set -u
TAG=${1:-r01}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"

# 2. per-kernel time of the same command (no CPU leg: it is not GPU work)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
    python bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv

# 3. HBM traffic counters, one pass each
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- \
      python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1
  find $OUT/${TAG}_pmc_$C -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_pmc_$C.csv
done
python tools/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv > $OUT/${TAG}_pmc_hbm.json
cat $OUT/${TAG}_pmc_hbm.json | head -50
cp $OUT/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json  # bench.py quotes roofline.traffic from the committed path

# 1. the default bench line (includes the bounded CPU baseline); after the PMC summary so that roofline.traffic is this run's
timeout 600 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench_line.json

# 4. recurrent (cfg4 shape) update micro-benchmark + its kernel stats
timeout 600 python benchmarks/rnn_update_bench.py > $OUT/${TAG}_rnn_update_line.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rnn_stats -- \
    python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 > $OUT/${TAG}_rnn_stats.log 2>&1
find $OUT/${TAG}_rnn_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_rnn_update_kernel_stats.csv
cat $OUT/${TAG}_rnn_update_line.json
# keep the merge small: the raw rocprof trees are not needed
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_rnn_stats

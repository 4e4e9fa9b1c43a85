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
    python bench.py --no-cpu-baseline --no-other-configs > $OUT/${TAG}_stats.log 2>&1
find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv

# 3. HBM traffic counters, one pass each
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- \
      python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_pmc_$C.log 2>&1
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
# 5. the other BASELINE configurations / shapes (builder-run lines beside the headline) + their kernel stats
stats() {  # stats <name> <command...>: kernel-trace summary of a command -> $OUT/${TAG}_<name>_kernel_stats.csv
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st_$name -- "$@" > $OUT/${TAG}_st_$name.log 2>&1
  find $OUT/${TAG}_st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_st_$name
}
timeout 600 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
stats cfg4_mpe python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2
timeout 600 python benchmarks/cfg5_ttt_bench.py > $OUT/${TAG}_cfg5_ttt_line.json 2>/dev/null
stats cfg5_ttt python benchmarks/cfg5_ttt_bench.py --steps 4 --warmup 2
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_rollout > $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_reset >> $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
timeout 600 python benchmarks/shape_sweep.py > $OUT/${TAG}_shape_sweep.jsonl 2>/dev/null
stats cfg3_shape python benchmarks/shape_sweep.py --only cfg3_halfcheetah_shape
timeout 600 python benchmarks/host_env_bench.py > $OUT/${TAG}_host_env_line.json 2>/dev/null
# 5b. the general tower path (non-default towers): bench lines + kernel stats at hidden 128
(python benchmarks/generic_bench.py --steps 5 --warmup 3; python benchmarks/generic_bench.py --steps 3 --warmup 3 --layer_N 4
 python benchmarks/generic_bench.py --steps 3 --warmup 3 --hidden_size 256; python benchmarks/generic_bench.py --steps 3 --warmup 3 --share) \
  2>/dev/null | grep generic_tower_path > $OUT/${TAG}_generic_lines.jsonl
stats generic_h128 python benchmarks/generic_bench.py --steps 3 --warmup 3
stats generic_recurrent_h128 python benchmarks/cfg4_mpe_bench.py --hidden_size 128 --steps 3 --warmup 3
(python benchmarks/cfg4_mpe_bench.py --hidden_size 128 --steps 5 --warmup 3; python benchmarks/cfg4_mpe_bench.py --hidden_size 256 --steps 3 --warmup 3
 python benchmarks/cfg4_mpe_bench.py --hidden_size 64 --layer_N 2 --steps 5 --warmup 3) 2>/dev/null | grep cfg4_mpe > $OUT/${TAG}_generic_recurrent_lines.jsonl
# 6. the per-rank shard of the strong-scaling bench at 8 and 2 GPUs (512 / 2048 of the 4096 envs), on one GPU
for E in 512 2048; do
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --envs $E > $OUT/${TAG}_bench_envs${E}_line.json 2>/dev/null
done
stats bench_envs512 python bench.py --no-cpu-baseline --no-other-configs --envs 512
# 7. issue / wait counters of the tower kernel
bash tools/pmc_tower.sh > $OUT/${TAG}_pmc_tower.txt 2>&1
# 8. the GPU test suite
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
tail -3 $OUT/${TAG}_pytest_gpu.log
# keep the merge small: the raw rocprof trees are not needed
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_rnn_stats

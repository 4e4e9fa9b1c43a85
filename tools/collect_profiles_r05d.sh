#!/usr/bin/env bash
# Round 5, re-collection after the cooperative recurrent rollout (cfg4 lines and kernel statistics) + the GPU test suite
set -u
export ORL_KEEP_BUILD=1
TAG=r05
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st_$name -- "$@" > $OUT/${TAG}_st_$name.log 2>&1
  find $OUT/${TAG}_st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_st_$name
}
timeout 600 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
stats cfg4_mpe python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2
(for g in fp32 fp32_recompute split; do python benchmarks/rnn_update_bench.py --tower-gemm $g; done) > $OUT/${TAG}_rnn_update_lines.jsonl 2>/dev/null
stats rnn_update python benchmarks/rnn_update_bench.py --iters 3 --warmup 1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
tail -3 $OUT/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
head -c 600 $OUT/${TAG}_cfg4_mpe_line.json; echo
head -5 $OUT/${TAG}_cfg4_mpe_kernel_stats.csv

#!/usr/bin/env bash
# Round 5, the lines of the FINAL tree: bench command (kernel statistics + the default line) and the cfg4 / recurrent-update lines
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
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq|Wavefront Size" | tail -8; rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -30) > $OUT/${TAG}_box_facts.txt 2>&1
stats bench python bench.py --no-cpu-baseline --no-other-configs
timeout 600 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench_line.json; echo
timeout 600 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
stats cfg4_mpe python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2
(for g in fp32 fp32_recompute split; do python benchmarks/rnn_update_bench.py --tower-gemm $g; done) > $OUT/${TAG}_rnn_update_lines.jsonl 2>/dev/null
stats rnn_update python benchmarks/rnn_update_bench.py --iters 3 --warmup 1
timeout 600 python -m pytest tests/test_mpe_gpu.py -m gpu -q 2>&1 | tail -3
head -4 $OUT/${TAG}_bench_kernel_stats.csv
head -4 $OUT/${TAG}_cfg4_mpe_kernel_stats.csv
cat $OUT/${TAG}_cfg4_mpe_line.json
cat $OUT/${TAG}_box_facts.txt

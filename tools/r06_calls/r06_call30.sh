#!/usr/bin/env bash
# round 6, call 30: the recurrent wgrad kernel (HBM bound on the tape at 3.9 TB/s) with 3 of its 6 bf16 products and no third
# split term (-DORL_SPLIT_PROBE: wrong numerics, timing only) - would a two-term split let it run closer to the HBM rate?
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default rnnwprobe default rnnwprobe; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"
  rm -rf gpurun_out/c30; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c30 -- python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 > gpurun_out/c30.log 2>&1
  f=$(find gpurun_out/c30 -name "*kernel_stats.csv" | head -1)
  head -4 "$f" | sed 's/(.*)//' | cut -c1-120
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

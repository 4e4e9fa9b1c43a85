#!/usr/bin/env bash
# usage: r05_call22.sh <variant...> : kernel time of the fused recurrent rollout per build variant (cfg4), tests on `coop`
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c22
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/coop.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_mpe_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_mpe.log
: > $OUT/kstat.txt
for v in "$@"; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v" >> $OUT/kstat.txt
  bash tools/kstat.sh c4_${v}_$RANDOM python benchmarks/cfg4_mpe_bench.py --steps 3 --warmup 1 2>&1 | grep rollout >> $OUT/kstat.txt
done
cat $OUT/kstat.txt
cp variants/coop.so openrl_amd/csrc/liborl_hip.so

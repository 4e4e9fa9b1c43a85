#!/usr/bin/env bash
# adv_normalize_pack with the smaller tile for wide observations (pack1) against before (pack0): the recurrent training goldens on pack1, kernel stats of cfg4
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c27
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/pack1.so openrl_amd/csrc/liborl_hip.so
timeout 600 python -m pytest tests/test_rnn_train_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for v in pack0 pack1; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; bash tools/kstat.sh c4_$v python benchmarks/cfg4_mpe_bench.py --steps 6 --warmup 2 2>&1 | grep -i "pack\|rollout"
done | tee $OUT/kstat.txt
cp variants/pack1.so openrl_amd/csrc/liborl_hip.so

#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c20
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/red1.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rnn_kernels_gpu.py tests/test_rnn_train_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_rnn.log
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in red2 red1; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python benchmarks/rnn_update_bench.py --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'rnn ms_per_epoch', round(r['ms_per_epoch'],4))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
cp variants/red1.so openrl_amd/csrc/liborl_hip.so

#!/usr/bin/env bash
# tape row rotation g & 7 (rot8) against 4 (g & 3) (rot4): recurrent tests on rot8, epoch time alternated, kernel stats
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/rot8.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rnn_kernels_gpu.py tests/test_rnn_train_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_rnn.log
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in rot4 rot8; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python benchmarks/rnn_update_bench.py --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'rnn ms_per_epoch', round(r['ms_per_epoch'],4))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
for v in rot4 rot8; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; bash tools/kstat.sh rnn_$v python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 2>&1 | head -3
done
cp variants/rot8.so openrl_amd/csrc/liborl_hip.so

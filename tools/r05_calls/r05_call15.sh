#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for rep in 1 2; do
  for v in default rp120 rp124 rp136; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python benchmarks/rnn_update_bench.py --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'rnn ms_per_epoch', round(r['ms_per_epoch'],4))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so

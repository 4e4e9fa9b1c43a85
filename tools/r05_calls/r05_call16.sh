#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in default iglp0 iglp1; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so

#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c18
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in default dmalate; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python benchmarks/shape_sweep.py --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: continue
    print('$v', r.get('bench'), 'pair_ms', r.get('tower_pair_ms'), 'iter_ms', r.get('ms_per_iteration'))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
cp variants/dmalate.so openrl_amd/csrc/liborl_hip.so
timeout 600 python -m pytest tests/test_ppo_update_gpu.py -m gpu -x -q 2>&1 | tail -2
cp variants/default.so openrl_amd/csrc/liborl_hip.so

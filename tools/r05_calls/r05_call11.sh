#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab_tower.txt
for rep in 1 2; do
  for v in default tr8 w10 w12 w12s; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'], 'frac', r['roofline']['frac'])" >> $OUT/ab_tower.txt
  done
done
cat $OUT/ab_tower.txt
for v in w12; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  timeout 300 python -m pytest tests/test_ppo_update_gpu.py -m gpu -x -q -k "full_size_update_matches or train_matches" 2>&1 | tail -3
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

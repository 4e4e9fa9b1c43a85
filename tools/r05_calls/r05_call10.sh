#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab_tower.txt
for rep in 1 2 3 4; do
  for v in default setprio1 setprio2; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'], 'frac', r['roofline']['frac'])" >> $OUT/ab_tower.txt
  done
done
cat $OUT/ab_tower.txt
for v in default setprio2; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  python benchmarks/shape_sweep.py --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: continue
    print('$v', r.get('bench'), 'pair_ms', r.get('tower_pair_ms'), 'iter_ms', r.get('ms_per_iteration'))" | tee -a $OUT/ab_tower.txt
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

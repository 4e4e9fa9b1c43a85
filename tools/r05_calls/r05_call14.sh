#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ppo_update_gpu.py tests/test_a2c.py tests/test_multiagent_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in epi0 default; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab.txt
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'envs512 ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab.txt
  done
done
for v in epi0 default; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  python benchmarks/shape_sweep.py --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: continue
    print('$v', r.get('bench'), 'pair_ms', r.get('tower_pair_ms'), 'iter_ms', r.get('ms_per_iteration'))" >> $OUT/ab.txt
done
cat $OUT/ab.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so

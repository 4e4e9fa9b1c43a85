#!/usr/bin/env bash
# Round 5, GPU call 2: the split forms (and/sub, fixed dot2c, packed sub), one wave per SIMD, phase profiles old / new, parity.
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 tools/split_dot2c_probe.hip -o /tmp/split_probe > /dev/null 2>&1 && /tmp/split_probe > $OUT/split_dot2c_probe.txt 2>&1
tail -3 $OUT/split_dot2c_probe.txt
: > $OUT/ab_tower.txt
for rep in 1 2 3; do
  for v in r04like default form1 form2 w4; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'], 'frac', r['roofline']['frac'])" >> $OUT/ab_tower.txt
  done
done
cat $OUT/ab_tower.txt
for v in prof_old prof_new; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v" >> $OUT/tower_phase_prof.txt
  python tools/tower_phase_prof.py 2>/dev/null >> $OUT/tower_phase_prof.txt
done
cat $OUT/tower_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ppo_update_gpu.py tests/test_gen_tower_gpu.py tests/test_kernels_gpu.py tests/test_a2c.py -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $OUT/pytest_subset.log

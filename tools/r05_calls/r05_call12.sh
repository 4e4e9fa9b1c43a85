#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c12
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/ab.txt
for rep in 1 2 3; do
  for v in default ln1p fc2c ln1p_fc2; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab.txt
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'envs512 ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
for v in default ln1p_fc2; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  python benchmarks/rnn_update_bench.py --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'rnn ms_per_epoch', round(r['ms_per_epoch'],4))" | tee -a $OUT/ab.txt
  python benchmarks/cfg4_mpe_bench.py 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'cfg4 iter', round(r['ms_per_iteration'],3), 'rollout', round(r['ms_rollout'],4))" | tee -a $OUT/ab.txt
  python benchmarks/shape_sweep.py --steps 5 --warmup 2 --only cfg3_halfcheetah_shape 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'cfg3 iter', r['ms_per_iteration'], 'pair', r['tower_pair_ms'])" | tee -a $OUT/ab.txt
done
cp variants/ln1p_fc2.so openrl_amd/csrc/liborl_hip.so
timeout 1200 python -m pytest tests -m gpu -q -x -k "not learning" 2>&1 | tail -6 | tee $OUT/pytest_ln1p.log
cp variants/default.so openrl_amd/csrc/liborl_hip.so

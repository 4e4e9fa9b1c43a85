#!/usr/bin/env bash
# round 6, call 7: orl_ppo_step (one-launch optimiser step) - bit-for-bit tests, multi-rank cases, A/B against two launches
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ppo_update_gpu.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-250
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-250
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_max'), d['roofline']['launch_ms'])"; }
for i in 1 2 3; do
for k in step two_launch; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --optim-step $k 2>/dev/null | line "$k 4096"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --optim-step $k --envs 512 2>/dev/null | line "$k 512"
done
done
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -- "$@" > /dev/null 2>&1
  find $OUT/st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/r06_${name}_kernel_stats.csv
  rm -rf $OUT/st_$name
  head -5 $OUT/r06_${name}_kernel_stats.csv | cut -c1-150
}
stats step_bench python bench.py --no-cpu-baseline --no-other-configs
stats step_bench_envs512 python bench.py --no-cpu-baseline --no-other-configs --envs 512

#!/usr/bin/env bash
# round 6, call 11: the speculative rollout kernel - tests, then rollout time and bench lines against the unspeculated chain
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rollout_gpu.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-250
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_max'), d['roofline']['launch_ms'])"; }
for i in 1 2; do
for k in chain chain_nospec; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k 2>/dev/null | line "$k 4096"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --envs 512 2>/dev/null | line "$k 512"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --env cartpole 2>/dev/null | line "$k cartpole 4096"
done
done
for e in synthetic cartpole; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_sp -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --env $e > /dev/null 2>&1
  f=$(find $OUT/st_sp -name '*kernel_stats.csv' | head -1); echo "$e $(grep rollout $f | sed 's/(orl::RolloutArgs)//' | cut -c1-120)"; rm -rf $OUT/st_sp
done

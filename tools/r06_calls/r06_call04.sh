#!/usr/bin/env bash
# round 6, call 4: the chain rollout kernel - tests, then bench lines chain vs lockstep at 4096 / 512 envs and on CartPole physics
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rollout_gpu.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-250
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['launch_ms'])"; }
for i in 1 2; do
  for k in chain lockstep; do
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k 2>/dev/null | line "$k 4096"
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --envs 512 2>/dev/null | line "$k 512"
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --env cartpole 2>/dev/null | line "$k cartpole 4096"
  done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_chain -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 > /dev/null 2>&1
find $OUT/st_chain -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/r06_chain_kernel_stats.csv; rm -rf $OUT/st_chain
head -12 $OUT/r06_chain_kernel_stats.csv | cut -c1-200

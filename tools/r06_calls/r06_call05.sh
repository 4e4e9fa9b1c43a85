#!/usr/bin/env bash
# round 6, call 5: chain rollout - tests, phase profile (variants/prof.so), bench lines and kernel statistics
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rollout_gpu.py -m gpu -q 2>&1 | tail -25 | cut -c1-250
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
timeout 300 python tools/rollout2_phase_prof.py 2>/dev/null | grep -v "^{" > $OUT/r06_rollout2_phase_prof.txt
timeout 300 python tools/rollout2_phase_prof.py --env cartpole 2>/dev/null | grep -v "^{" >> $OUT/r06_rollout2_phase_prof.txt
cat $OUT/r06_rollout2_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['launch_ms'])"; }
for k in chain lockstep; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k 2>/dev/null | line "$k 4096"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --envs 512 2>/dev/null | line "$k 512"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --env cartpole 2>/dev/null | line "$k cartpole 4096"
done
for e in 4096 512; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_chain -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --envs $e > /dev/null 2>&1
find $OUT/st_chain -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/r06_chain_${e}_kernel_stats.csv; rm -rf $OUT/st_chain
head -6 $OUT/r06_chain_${e}_kernel_stats.csv | cut -c1-160
done

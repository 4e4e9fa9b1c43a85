#!/usr/bin/env bash
# round 6, call 6: the whole GPU suite on the shipped build, the comparison kernels' tests on the experimental build, the chain
# rollout's phase profile, bench lines (chain / lockstep rollout at 4096 and 512 envs, CartPole physics) and kernel statistics
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 | cut -c1-250 > $OUT/r06_pytest_gpu.log
tail -12 $OUT/r06_pytest_gpu.log
cp variants/experiments.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rnn_train_gpu.py tests/test_ppo_update_gpu.py -m gpu -q \
  -k "split or fp32 or one_launch or two_image" 2>&1 | tail -15 | cut -c1-250 > $OUT/r06_pytest_gpu_experiments.log
tail -6 $OUT/r06_pytest_gpu_experiments.log
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
timeout 300 python tools/rollout2_phase_prof.py 2>/dev/null | grep -v "^{" > $OUT/r06_rollout2_phase_prof.txt
timeout 300 python tools/rollout2_phase_prof.py --env cartpole 2>/dev/null | grep -v "^{" >> $OUT/r06_rollout2_phase_prof.txt
cat $OUT/r06_rollout2_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_max'), d['roofline']['launch_ms'])"; }
for i in 1 2; do
for k in chain lockstep; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k 2>/dev/null | line "$k 4096"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --envs 512 2>/dev/null | line "$k 512"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel $k --env cartpole 2>/dev/null | line "$k cartpole 4096"
done
done
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -- "$@" > /dev/null 2>&1
  find $OUT/st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/r06_${name}_kernel_stats.csv
  rm -rf $OUT/st_$name
  head -4 $OUT/r06_${name}_kernel_stats.csv | cut -c1-150
}
stats bench python bench.py --no-cpu-baseline --no-other-configs
stats bench_envs512 python bench.py --no-cpu-baseline --no-other-configs --envs 512
stats bench_cartpole python bench.py --no-cpu-baseline --no-other-configs --env cartpole

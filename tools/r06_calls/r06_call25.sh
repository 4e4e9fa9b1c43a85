#!/usr/bin/env bash
# round 6, call 25: the chain rollout's policy fc2 on the fp16 split (6 MFMAs + 32 VALU instead of 16 fp32 MFMAs per step and wave)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_rollout_gpu.py tests/test_ttt_gpu.py tests/test_kernels_gpu.py tests/test_layernorm_adversarial_gpu.py tests/test_examples_gpu.py tests/test_learning_gpu.py -m gpu -q 2>&1 | tail -15 | cut -c1-300
for v in default prev default prev; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"
  for e in 4096 512; do
  rm -rf gpurun_out/c25; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c25 -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --envs $e > gpurun_out/c25.log 2>&1
  f=$(find gpurun_out/c25 -name "*kernel_stats.csv" | head -1)
  tail -1 gpurun_out/c25.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  grep rollout2 "$f" | sed 's/(.*)//' | cut -c1-120
  done
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

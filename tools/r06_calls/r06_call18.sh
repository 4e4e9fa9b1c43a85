#!/usr/bin/env bash
# round 6, call 18: accuracy of the fp16 two-term towers against float64 at full size, beside the bf16 three-term build and the
# fp32 MFMA (ORL_BUILD_EXPERIMENTS libraries); then the whole GPU suite on the shipped build
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in experiments experiments_bf16x3; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 900 python -m pytest tests/test_ppo_update_gpu.py -m gpu -q -s -k "as_accurate" 2>&1 | grep -E "tower:|passed|failed" | cut -c1-300
done
cp variants/experiments.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests -m gpu -q -k "split or fp32 or one_launch or two_image" 2>&1 | tail -6 | cut -c1-300
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 | cut -c1-400 | tee gpurun_out/r06_pytest_gpu_f16.log

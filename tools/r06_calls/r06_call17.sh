#!/usr/bin/env bash
# round 6, call 17: the towers' 64-wide GEMMs as two-term fp16 splits (3 products, power-of-two operand scaling) - the split's
# own probe, parity of the update / rollout suites, and the bench against the three-term bf16 build (variants/bf16x3.so)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
variants/bin/f16test | tail -3
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_ppo_update_gpu.py tests/test_kernels_gpu.py tests/test_rollout_gpu.py tests/test_layernorm_adversarial_gpu.py tests/test_ttt_gpu.py tests/test_reference_style_gpu.py -m gpu -q 2>&1 | tail -25 | cut -c1-400
for v in default bf16x3 default bf16x3; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"
done

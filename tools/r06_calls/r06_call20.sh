#!/usr/bin/env bash
# round 6, call 20: one running gradient scale per wave (folded into rstd2; xhat1 unscaled in the wgrad) against the first fp16
# build (per-tile scale: variants/f16v1.so) - parity of the update suites, then the bench in alternation
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_ppo_update_gpu.py tests/test_split_scaling_gpu.py tests/test_layernorm_adversarial_gpu.py tests/test_kernels_gpu.py tests/test_reference_style_gpu.py tests/test_learning_gpu.py -m gpu -q 2>&1 | tail -12 | cut -c1-300
for v in default f16v1 default f16v1; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

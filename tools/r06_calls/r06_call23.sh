#!/usr/bin/env bash
# round 6, call 23: the record-DMA wait as a dependency of the ring pointer (small-observation build) - parity of the update suites
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_ppo_update_gpu.py tests/test_split_scaling_gpu.py tests/test_layernorm_adversarial_gpu.py tests/test_reference_style_gpu.py tests/test_learning_gpu.py tests/test_multirank_gpu.py tests/test_examples_gpu.py -m gpu -q 2>&1 | tail -8 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['frac'], d['roofline']['launch_ms'])"

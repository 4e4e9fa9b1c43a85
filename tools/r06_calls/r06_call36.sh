#!/usr/bin/env bash
# round 6, call 36: ONE running scale of dz2 for dgrad and wgrad, applied by explicit v_ldexp (xhat1 unscaled in the wgrad:
# - 18 VALU per tile; variants/runscale.so = -DORL_TOWER_RUNSCALE=1) against the per-tile scale (shipped): parity, then the bench
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/runscale.so openrl_amd/csrc/liborl_hip.so
timeout 1800 python -m pytest tests/test_ppo_update_gpu.py tests/test_split_scaling_gpu.py tests/test_layernorm_adversarial_gpu.py -m gpu -q 2>&1 | tail -4 | cut -c1-250
for v in default runscale default runscale; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['launch_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

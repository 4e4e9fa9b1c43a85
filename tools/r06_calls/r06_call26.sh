#!/usr/bin/env bash
# round 6, call 26: the tower pair's wave-wide sums of the epilogue and the staging's 8-lane sums on the VALU (DPP / permlane)
# instead of ds_bpermute butterflies - parity, the phase profile's prologue / epilogue, bench at 4096 and 512 envs against the
# previous build (variants/prev.so)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_ppo_update_gpu.py tests/test_split_scaling_gpu.py tests/test_kernels_gpu.py tests/test_rollout_gpu.py tests/test_reference_style_gpu.py tests/test_multirank_gpu.py -m gpu -q 2>&1 | tail -8 | cut -c1-300
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
python tools/tower_phase_prof.py 2>/dev/null | grep -E "per launch|epilogue|cycles per tile"
for v in default prev default prev; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"
  for e in 4096 512; do
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs --envs $e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['launch_ms'])"
  done
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

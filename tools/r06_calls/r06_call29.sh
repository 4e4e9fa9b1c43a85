#!/usr/bin/env bash
# round 6, call 29: the DMA wait's ring pointer laundered as an LDS pointer (no flat loads; default) against the first form
# (variants/prev.so), and the same wait on the wide builds too (variants/depall.so = -DORL_DMA_WAIT_DEP=2) at cfg3's / cfg5's shapes
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default prev depall default prev depall; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python benchmarks/shape_sweep.py --steps 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['bench'], d['ms_per_iteration'], d['tower_pair_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 1800 python -m pytest tests/test_ppo_update_gpu.py tests/test_split_scaling_gpu.py tests/test_reference_style_gpu.py -m gpu -q 2>&1 | tail -4 | cut -c1-200

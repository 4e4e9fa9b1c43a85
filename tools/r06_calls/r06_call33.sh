#!/usr/bin/env bash
# round 6, call 33: the policy / critic cost ratio that splits the CUs between the towers of a wide-head pair (ORL_PAIR_WP_WIDE:
# 1.10 = back to back above 24 576 tiles; 1.09 = side by side everywhere; 1.20 .. 1.50 = other splits for cfg3; variants/wpNNN.so =
# -DORL_PAIR_WP_WIDE=N.NN.  Run twice: {1.09, 1.20, 1.30} and {1.30, 1.40, 1.50}.  Result: 1.30 for wide Gaussian heads (ORL_PAIR_WP_WIDE_GAUSS))
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default wp130 wp140 wp150 default wp130 wp140 wp150; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"
  for s in cfg3 cfg5; do timeout 300 python benchmarks/shape_sweep.py --only $s --steps 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['bench'], d['ms_per_iteration'], d['tower_pair_ms'])"; done
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

#!/usr/bin/env bash
# round 6, call 10: what a second trunk group on the same four SIMDs costs the rollout chain (variants/ro2_shadow.so, timing only)
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default ro2_shadow default ro2_shadow; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_sh -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --envs 512 > /dev/null 2>&1
  f=$(find $OUT/st_sh -name '*kernel_stats.csv' | head -1); echo "$v $(grep rollout2 $f | sed 's/.*)",//' | cut -d, -f1-3)"; rm -rf $OUT/st_sh
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

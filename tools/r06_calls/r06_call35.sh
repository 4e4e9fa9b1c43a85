#!/usr/bin/env bash
# round 6, call 35: does the next epoch's permutation (extra workgroups of the apply launch) lengthen the optimiser step?  apply kernel
# time with --perm device (shipped) against --perm identity (no permutation job)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
for pm in device identity; do
  rm -rf gpurun_out/c35; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c35 -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --perm $pm > gpurun_out/c35.log 2>&1
  f=$(find gpurun_out/c35 -name "*kernel_stats.csv" | head -1)
  echo "== perm $pm"; grep -E "apply|reduce_pair|perm_feistel" "$f" | sed 's/(.*)//' | cut -c1-110
done

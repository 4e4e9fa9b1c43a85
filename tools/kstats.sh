#!/bin/bash
# usage: tools/kstats.sh NAME cmd...   - rocprofv3 --kernel-trace --stats of one command (bounded), the per-kernel
# summary copied to gpurun_out/NAME_kernel_stats.csv and its first lines printed.
NAME=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
D=/tmp/ks_$NAME
rm -rf $D
( cd $REPO && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- "$@" > $REPO/gpurun_out/${NAME}.log 2>&1 )
tail -1 $REPO/gpurun_out/${NAME}.log
F=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ] && [ -f "$F" ]; then
  cp "$F" $REPO/gpurun_out/${NAME}_kernel_stats.csv
  head -${KSTATS_LINES:-22} "$F" | cut -c1-220
else
  echo "kstats: no kernel_stats.csv under $D"; ls -R $D 2>/dev/null | head -20
fi

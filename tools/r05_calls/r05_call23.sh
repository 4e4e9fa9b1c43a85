#!/usr/bin/env bash
# usage: r05_call23.sh <variant...> : per-launch durations of the fused recurrent rollout over a longer run (cfg4, 14 rollouts)
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c23
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $OUT/launches.txt
for v in "$@"; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  d=$OUT/tr_$v
  rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python benchmarks/cfg4_mpe_bench.py --steps 12 --warmup 2 > $OUT/tr_$v.log 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" >> $OUT/launches.txt <<'PY'
import csv,sys
out=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in csv.DictReader(open(sys.argv[1])) if 'rollout' in r['Kernel_Name']]
s=sorted(out)
print(sys.argv[2], 'median', round(s[len(s)//2],1), 'min', round(s[0],1), [round(x) for x in out])
PY
  rm -rf $d
done
cat $OUT/launches.txt
cp variants/coop.so openrl_amd/csrc/liborl_hip.so

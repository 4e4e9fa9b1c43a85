#!/usr/bin/env bash
# cfg4 line, kernel statistics and update-epoch lines of the final tree (after the pack tile change)
set -u
export ORL_KEEP_BUILD=1
TAG=r05
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 300 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st_cfg4 -- python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2 > $OUT/${TAG}_st_cfg4.log 2>&1
find $OUT/${TAG}_st_cfg4 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_cfg4_mpe_kernel_stats.csv
rm -rf $OUT/${TAG}_st_cfg4
timeout 200 python benchmarks/rnn_update_bench.py > $OUT/${TAG}_rnn_update_default_line.json 2>/dev/null
cat $OUT/${TAG}_cfg4_mpe_line.json; head -8 $OUT/${TAG}_cfg4_mpe_kernel_stats.csv | cut -c1-150
python -c "import json; print(json.load(open('$OUT/${TAG}_rnn_update_default_line.json'))['ms_per_epoch'])"

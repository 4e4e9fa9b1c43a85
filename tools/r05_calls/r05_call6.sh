#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python -c "
import json
r=json.load(open('$OUT/bench_line.json'))
print(r['value'], r['ms_per_step'], r['roofline']['launch_ms'], r['roofline']['frac'])
for o in r.get('other_configs',[]): print(o.get('workload','')[:40], o.get('ms_per_iteration'), o.get('dominant_kernel_ms'), o.get('frac'))
"

#!/usr/bin/env bash
# Per-kernel time table of the default bench workload (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash tools/kernel_stats.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python bench.py --no-cpu-baseline --steps 10 > /tmp/ks.log 2>&1
python tools/kstats.py "$(find /tmp/ks -name '*kernel_stats.csv' | head -1)"

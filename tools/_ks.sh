cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python bench.py --no-cpu-baseline --steps 10 > /tmp/ks.log 2>&1
tail -n1 /tmp/ks.log | cut -c1-200
python tools/kstats.py $(find /tmp/ks -name '*kernel_stats.csv' | head -1)

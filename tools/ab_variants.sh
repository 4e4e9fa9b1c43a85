#!/usr/bin/env bash
# usage: ab.sh variant1 variant2 ... ; runs bench for each variants/<v>.so
export ORL_KEEP_BUILD=1
for v in "$@"; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  for k in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['roofline']['launch_ms'], r['roofline']['frac'])"
  done
done

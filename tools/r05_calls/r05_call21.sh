#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c21
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mpe_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_mpe.log
bash tools/kstat.sh cfg4_coop python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2 2>&1 | tee $OUT/kstat.txt
python benchmarks/cfg4_mpe_bench.py 2>/dev/null | tail -1 > $OUT/cfg4_line.json
python -c "import json; r=json.load(open('$OUT/cfg4_line.json')); print({k:r[k] for k in ('value','ms_per_step') if k in r})"

#!/usr/bin/env bash
# Round 5, GPU call 3: the register-resident L = 2 recurrent row kernel - parity tests, then the update micro-benchmark and the
# cfg4 end-to-end line against the recompute kernel (same box).
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_rnn_kernels_gpu.py tests/test_rnn_train_gpu.py -m gpu -x -q -k "not general and not gen_ and not share" 2>&1 | tail -15 > $OUT/pytest_rnn.log
tail -6 $OUT/pytest_rnn.log
(for g in fp32 fp32_recompute fp32 fp32_recompute; do python benchmarks/rnn_update_bench.py --tower-gemm $g --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$g', 'ms_per_epoch', round(r['ms_per_epoch'],4), 'ms_per_train', round(r['ms_per_train'],3))"; done) | tee $OUT/rnn_update.txt
bash tools/kstat.sh rnn_l2 python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 2>&1 | tee $OUT/rnn_update_kstat.txt
timeout 300 python benchmarks/cfg4_mpe_bench.py 2>/dev/null | tail -1 | cut -c1-600 | tee $OUT/cfg4_line.json

#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_rnn_kernels_gpu.py tests/test_rnn_train_gpu.py -m gpu -x -q -k "not general and not gen_ and not share" 2>&1 | tail -15 > $OUT/pytest_rnn.log
tail -4 $OUT/pytest_rnn.log
(for g in fp32 fp32_recompute fp32 fp32_recompute; do python benchmarks/rnn_update_bench.py --tower-gemm $g --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$g', 'ms_per_epoch', round(r['ms_per_epoch'],4), 'ms_per_train', round(r['ms_per_train'],3))"; done) | tee $OUT/rnn_update.txt
bash tools/kstat.sh rnn_l2 python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 2>&1 | tee $OUT/rnn_update_kstat.txt
cp variants/prof_rnn.so openrl_amd/csrc/liborl_hip.so
python tools/rnn_phase_prof.py fp32 2>/dev/null | grep -v "^{" | tee $OUT/rnn_l2_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so

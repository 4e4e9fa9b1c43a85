#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
rocm-smi --showmemuse --showclocks 2>/dev/null | head -20
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | head
for g in fp32 fp32_recompute fp32; do python benchmarks/rnn_update_bench.py --tower-gemm $g --iters 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$g', 'ms_per_epoch', round(r['ms_per_epoch'],4))"; done
bash tools/kstat.sh rnn13 python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 2>&1 | head -4

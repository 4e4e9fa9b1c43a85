#!/usr/bin/env bash
# round-4 GPU call 8: unified-phase streamed row kernel: parity, timings, phase profile
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp variants/r4f.so openrl_amd/csrc/liborl_hip.so
timeout 300 python -m pytest tests/test_rnn_train_gpu.py tests/test_rnn_shapes_gpu.py tests/test_mpe_gpu.py -q -x 2>&1 | tail -4
for g in split fp32 split_w4 split fp32 split_w4; do
  timeout 120 python benchmarks/rnn_update_bench.py --tower-gemm $g 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rnn_update', '$g', round(r['ms_per_epoch'],4))"
done
cp variants/prof5.so openrl_amd/csrc/liborl_hip.so
(for g in split split_w4; do python tools/rnn_phase_prof.py $g 2>&1 | grep -v "^{" ; done) > gpurun_out/r04_phase_prof5.txt 2>&1; cat gpurun_out/r04_phase_prof5.txt

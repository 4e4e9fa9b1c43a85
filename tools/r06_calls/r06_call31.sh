#!/usr/bin/env bash
# round 6, call 31: the L = 2 recurrent row kernel over two-term fp16 images of its seven matrices (ORL_RNN_L2_H2; variants/prev.so =
# the fp32-MFMA kernel) - parity of the recurrent suites, the epoch's time, kernel statistics
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests/test_rnn_kernels_gpu.py tests/test_rnn_train_gpu.py tests/test_rnn_shapes_gpu.py tests/test_mpe_gpu.py tests/test_layernorm_adversarial_gpu.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-300
for v in default prev default prev; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python benchmarks/rnn_update_bench.py --tower-gemm fp32 2>/dev/null | tail -1 | cut -c1-260
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
rm -rf gpurun_out/c31; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c31 -- python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 > gpurun_out/c31.log 2>&1
f=$(find gpurun_out/c31 -name "*kernel_stats.csv" | head -1); head -4 "$f" | sed 's/(.*)//' | cut -c1-120

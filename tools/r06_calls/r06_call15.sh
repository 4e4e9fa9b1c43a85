#!/usr/bin/env bash
# round 6, call 15: Gaussian constants from an LDS table + the sampler's two row maxima in one butterfly - parity, then the
# rollout kernel's time at cfg3's / cfg5's shapes
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_rollout_gpu.py tests/test_ttt_gpu.py tests/test_reference_style_gpu.py tests/test_layernorm_adversarial_gpu.py -m gpu -q -x 2>&1 | tail -6 | cut -c1-300
for s in cfg3 cfg5; do
  rm -rf gpurun_out/c15_$s
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c15_$s -- python benchmarks/shape_sweep.py --only $s --steps 10 > gpurun_out/c15_$s.log 2>&1
  f=$(find gpurun_out/c15_$s -name "*kernel_stats.csv" | head -1)
  echo "== $s"; tail -1 gpurun_out/c15_$s.log | cut -c1-300; head -4 "$f" | sed 's/(.*)//' | cut -c1-200
done
timeout 300 python benchmarks/cfg5_ttt_bench.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-330

#!/usr/bin/env bash
# round 6, call 34: the GPU suite on the round's last binary (shipped build, then the comparison kernels' tests on the
# ORL_BUILD_EXPERIMENTS build), smoke()
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_pytest_gpu.log
cp variants/experiments.so openrl_amd/csrc/liborl_hip.so
timeout 2400 python -m pytest tests -m gpu -q -s -k "split or fp32 or one_launch or two_image" 2>&1 | grep -E "tower:|passed|failed" | cut -c1-250 > gpurun_out/r06_pytest_gpu_experiments.log
cp variants/default.so openrl_amd/csrc/liborl_hip.so
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tail -3 gpurun_out/r06_pytest_gpu.log; tail -2 gpurun_out/r06_pytest_gpu_experiments.log

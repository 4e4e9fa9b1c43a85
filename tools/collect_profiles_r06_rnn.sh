#!/usr/bin/env bash
# The recurrent part of tools/collect_profiles_r06.sh alone (after the L = 2 row kernel moved to fp16 images): cfg4 end to end,
# the recurrent update's lines / kernel statistics / counters / phase profile, the GPU suite.  gpurun --timeout 3000 -- 'bash tools/collect_profiles_r06_rnn.sh'
set -u
export ORL_KEEP_BUILD=1
TAG=r06
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st_$name -- "$@" > $OUT/${TAG}_st_$name.log 2>&1
  find $OUT/${TAG}_st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_st_$name
}
timeout 600 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
stats cfg4_mpe python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2
(for g in fp32 fp32_recompute; do python benchmarks/rnn_update_bench.py --tower-gemm $g; done) > $OUT/${TAG}_rnn_update_lines.jsonl 2>/dev/null
if [ -f variants/rnnfp32.so ]; then  # the fp32-MFMA form of the L = 2 kernel (-DORL_RNN_L2_H2=0), same box
  cp variants/rnnfp32.so openrl_amd/csrc/liborl_hip.so
  python benchmarks/rnn_update_bench.py --tower-gemm fp32 2>/dev/null | sed 's/"bench": "rnn_update"/"bench": "rnn_update (ORL_RNN_L2_H2=0: fp32 MFMA rows)"/' >> $OUT/${TAG}_rnn_update_lines.jsonl
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
stats rnn_update python benchmarks/rnn_update_bench.py --iters 3 --warmup 1
stats rnn_update_recompute python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 --tower-gemm fp32_recompute
(bash tools/pmc_rnn_row.sh fp32; bash tools/pmc_rnn_row.sh fp32_recompute) > $OUT/${TAG}_pmc_rnn.txt 2>&1
(bash tools/pmc_rnn_hbm.sh fp32; bash tools/pmc_rnn_hbm.sh fp32_recompute) > $OUT/${TAG}_pmc_rnn_hbm.txt 2>&1
if [ -f variants/prof.so ]; then
  cp variants/prof.so openrl_amd/csrc/liborl_hip.so
  python tools/rnn_phase_prof.py fp32 2>/dev/null | grep -v "^{" > $OUT/${TAG}_rnn_phase_prof.txt
  python tools/rnn_phase_prof.py fp32_recompute 2>/dev/null | grep -v "^{" >> $OUT/${TAG}_rnn_phase_prof.txt
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
if [ -f variants/experiments.so ]; then
  cp variants/experiments.so openrl_amd/csrc/liborl_hip.so
  timeout 2400 python -m pytest tests -m gpu -q -s -k "split or fp32 or one_launch or two_image" 2>&1 | grep -E "tower:|passed|failed" | cut -c1-250 > $OUT/${TAG}_pytest_gpu_experiments.log
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
tail -3 $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_rnn_update_lines.jsonl | cut -c1-200
rm -rf $OUT/pmcrow_* $OUT/pmct_* $OUT/pmcgth_*

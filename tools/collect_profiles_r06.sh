#!/usr/bin/env bash
# Round-6 measurements on one MI355X box (through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles_r06.sh'
# Everything lands under gpurun_out/r06_*; copy what is to be judged into profiles/ (tracked).  Counter passes are separate
# runs with --pmc only (no sys / hip / hsa traces), as the pool requires.  variants/prof.so = the ORL_PROF build of the tree.
set -u
export ORL_KEEP_BUILD=1
TAG=r06
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
stats() {  # stats <name> <command...>: rocprofv3 kernel-trace summary of a command -> $OUT/${TAG}_<name>_kernel_stats.csv
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st_$name -- "$@" > $OUT/${TAG}_st_$name.log 2>&1
  find $OUT/${TAG}_st_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_st_$name
}
# 1. per-kernel time of the bench command (no CPU leg: it is not GPU work)
stats bench python bench.py --no-cpu-baseline --no-other-configs
# 2. HBM traffic counters, one pass each
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- \
      python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_pmc_$C.log 2>&1
  find $OUT/${TAG}_pmc_$C -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/${TAG}_pmc_$C.csv
  rm -rf $OUT/${TAG}_pmc_$C
done
python tools/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv > $OUT/${TAG}_pmc_hbm.json
cp $OUT/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json  # bench.py quotes roofline.traffic from the committed path
# 3. issue / wait / LDS counters of the tower kernel (bench.py reads the busy fractions from the committed path)
bash tools/pmc_tower.sh > $OUT/${TAG}_pmc_tower.txt 2>&1
cp $OUT/${TAG}_pmc_tower.txt profiles/${TAG}_pmc_tower.txt
# 4. the default bench line (bounded CPU baseline + other_configs), after the PMC summaries
timeout 600 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench_line.json; echo
# 5. per-rank shards of the strong-scaling bench (512 / 2048 of the 4096 envs) on one GPU, and a 2-rank run on ONE GPU
for E in 512 2048; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --envs $E > $OUT/${TAG}_bench_envs${E}_line.json 2>/dev/null
done
stats bench_envs512 python bench.py --no-cpu-baseline --no-other-configs --envs 512
# 5b. configs[1] on the device CartPole-v1 physics (line + kernel statistics: VERDICT r5 item 6), and the round-5 lock-step
#     rollout kernel on both envs for comparison
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --env cartpole > $OUT/${TAG}_bench_cartpole_line.json 2>/dev/null
stats bench_cartpole python bench.py --no-cpu-baseline --no-other-configs --env cartpole
(timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel lockstep; timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel lockstep --env cartpole; timeout 300 python bench.py --no-cpu-baseline --no-other-configs --rollout-kernel lockstep --envs 512) > $OUT/${TAG}_bench_lockstep_lines.jsonl 2>/dev/null
ORL_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_2ranks_one_gpu_line.json 2>/dev/null
# 6. other BASELINE shapes: tower-pair times, cfg4 / cfg5 end to end
python benchmarks/shape_sweep.py > $OUT/${TAG}_shape_sweep.jsonl 2>/dev/null
stats cfg3_shape python benchmarks/shape_sweep.py --only cfg3_halfcheetah_shape
timeout 600 python benchmarks/cfg4_mpe_bench.py > $OUT/${TAG}_cfg4_mpe_line.json 2>/dev/null
stats cfg4_mpe python benchmarks/cfg4_mpe_bench.py --steps 4 --warmup 2
timeout 600 python benchmarks/cfg5_ttt_bench.py > $OUT/${TAG}_cfg5_ttt_line.json 2>/dev/null
stats cfg5_ttt python benchmarks/cfg5_ttt_bench.py --steps 4 --warmup 2
timeout 600 python benchmarks/cfg5_ttt_bench.py --rollout-kernel lockstep > $OUT/${TAG}_cfg5_ttt_lockstep_line.json 2>/dev/null
stats cfg5_ttt_lockstep python benchmarks/cfg5_ttt_bench.py --steps 4 --warmup 2 --rollout-kernel lockstep
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_rollout > $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_reset >> $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
timeout 600 python benchmarks/host_env_bench.py > $OUT/${TAG}_host_env_line.json 2>/dev/null
# 7. the recurrent update: the row kernels (L = 2 register-resident default, recompute), kernel stats, counters
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
# 8. the general tower path (hidden 128): lines only (capped since round 4)
(python benchmarks/generic_bench.py --steps 5 --warmup 3; python benchmarks/generic_bench.py --steps 3 --warmup 3 --share) \
  2>/dev/null | grep generic_tower_path > $OUT/${TAG}_generic_lines.jsonl
# 9. phase profiles (timing build)
if [ -f variants/prof.so ]; then
  cp variants/prof.so openrl_amd/csrc/liborl_hip.so
  python tools/tower_phase_prof.py 2>/dev/null > $OUT/${TAG}_tower_phase_prof.txt
  python tools/tower_phase_prof.py --obs 18 --act 9 --T 200 2>/dev/null >> $OUT/${TAG}_tower_phase_prof.txt
  python tools/rnn_phase_prof.py fp32 2>/dev/null | grep -v "^{" > $OUT/${TAG}_rnn_phase_prof.txt
  python tools/rnn_phase_prof.py fp32_recompute 2>/dev/null | grep -v "^{" >> $OUT/${TAG}_rnn_phase_prof.txt
  python tools/rollout2_phase_prof.py 2>/dev/null | grep -v "^{" > $OUT/${TAG}_rollout2_phase_prof.txt
  python tools/rollout2_phase_prof.py --env cartpole 2>/dev/null | grep -v "^{" >> $OUT/${TAG}_rollout2_phase_prof.txt
  for shp in cfg3 cfg5; do
    echo "== shape $shp (benchmarks/shape_sweep.py)" >> $OUT/${TAG}_rollout2_phase_prof.txt
    python tools/rollout2_phase_prof.py --shape $shp 2>/dev/null | grep -v "^{" >> $OUT/${TAG}_rollout2_phase_prof.txt
  done
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
# 10. the GPU test suite
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
# 11. the comparison kernels' tests and the split's accuracy against float64 on the ORL_BUILD_EXPERIMENTS library
if [ -f variants/experiments.so ]; then
  cp variants/experiments.so openrl_amd/csrc/liborl_hip.so
  timeout 2400 python -m pytest tests -m gpu -q -s -k "split or fp32 or one_launch or two_image" 2>&1 | grep -E "tower:|passed|failed" | cut -c1-250 > $OUT/${TAG}_pytest_gpu_experiments.log
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
# 12. the split's own probe (one wave against float64)
[ -x variants/bin/sfg ] && variants/bin/sfg > $OUT/${TAG}_split_f16_gemm.txt 2>&1
tail -3 $OUT/${TAG}_pytest_gpu.log
rm -rf $OUT/pmcrow_* $OUT/pmct_* $OUT/pmcgth_*

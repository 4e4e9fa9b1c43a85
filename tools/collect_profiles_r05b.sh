#!/usr/bin/env bash
# Round 5, partial re-collection after the last tower change (wgrad fragments in halves): the lines and kernel stats it moves.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r05b.sh'
set -u
export ORL_KEEP_BUILD=1
TAG=r05
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
stats bench python bench.py --no-cpu-baseline --no-other-configs
timeout 600 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
for E in 512 2048; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --envs $E > $OUT/${TAG}_bench_envs${E}_line.json 2>/dev/null
done
python benchmarks/shape_sweep.py > $OUT/${TAG}_shape_sweep.jsonl 2>/dev/null
stats cfg3_shape python benchmarks/shape_sweep.py --only cfg3_halfcheetah_shape
timeout 600 python benchmarks/cfg5_ttt_bench.py > $OUT/${TAG}_cfg5_ttt_line.json 2>/dev/null
stats cfg5_ttt python benchmarks/cfg5_ttt_bench.py --steps 4 --warmup 2
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_rollout > $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
timeout 600 python benchmarks/cfg5_ttt_bench.py --opponent pool --sampling per_reset >> $OUT/${TAG}_cfg5_selfplay_line.json 2>/dev/null
if [ -f variants/prof.so ]; then
  cp variants/prof.so openrl_amd/csrc/liborl_hip.so
  python tools/tower_phase_prof.py 2>/dev/null > $OUT/${TAG}_tower_phase_prof.txt
  python tools/tower_phase_prof.py --obs 18 --act 9 --T 200 2>/dev/null >> $OUT/${TAG}_tower_phase_prof.txt
  python tools/tower_phase_prof.py --obs 17 --box 6 --envs 1024 --T 200 2>/dev/null >> $OUT/${TAG}_tower_phase_prof.txt
  cp variants/default.so openrl_amd/csrc/liborl_hip.so
fi
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/${TAG}_pytest_gpu.log
tail -3 $OUT/${TAG}_pytest_gpu.log

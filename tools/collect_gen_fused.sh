#!/usr/bin/env bash
# Round-3 measurements of the cross-layer fused general towers, one gpurun call:
#   bash tools/collect_gen_fused.sh TAG      -> gpurun_out/TAG_*  (copy what is to be judged into profiles/)
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/${TAG}_generic_fused_lines.jsonl
for args in "--hidden_size 128" "--hidden_size 128 --gen_update layerwise" "--hidden_size 64 --layer_N 2 --activation_id 0" "--hidden_size 128 --layer_N 2" "--hidden_size 128 --obs_dim 17" "--hidden_size 64 --share"; do
  timeout 120 python benchmarks/generic_bench.py $args 2>/dev/null | grep generic_tower_path >> $OUT/${TAG}_generic_fused_lines.jsonl
done
cat $OUT/${TAG}_generic_fused_lines.jsonl
# the one-launch update (orl_gt_train), both towers on ONE stream so that the per-kernel durations add up
KSTATS_LINES=12 timeout 330 tools/kstats.sh ${TAG}_generic_train_h128 python benchmarks/generic_bench.py --hidden_size 128 --steps 3 --warmup 1 --one_stream

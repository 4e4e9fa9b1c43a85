#!/usr/bin/env bash
# A/B of the tower-kernel variants of round 4 on one box (through gpurun): for each build in variants/ and each run-time GEMM
# path, the cfg3 / cfg5 shapes of benchmarks/shape_sweep.py; prints one JSON line per run into gpurun_out/r04_ab_tower.jsonl
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r04_ab_tower.jsonl
: > $OUT
for v in "$@"; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  for g in split split_two_image; do
    for rep in 1 2; do
      for shp in cfg3 cfg5; do
        python benchmarks/shape_sweep.py --only $shp --steps 5 --warmup 2 --tower-gemm $g 2>/dev/null | tail -1 | \
          python -c "import sys,json; r=json.loads(sys.stdin.read()); r['build']='$v'; print(json.dumps(r))" >> $OUT
      done
    done
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_ab_tower.jsonl"):
    r = json.loads(ln)
    print(r["build"], r["tower_gemm"], r["bench"], r["tower_pair_ms"], r["tower_pair_frac_of_fp32_mfma_peak"], r["ms_per_iteration"])
PY

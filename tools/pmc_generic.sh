#!/usr/bin/env bash
# HBM traffic of the general tower path's layer kernels (hidden 128, config-2 shape): FETCH_SIZE / WRITE_SIZE in their
# own rocprofv3 --pmc passes (no trace domains), per-kernel means printed as JSON lines.
#   gpurun --timeout 900 -- 'bash tools/pmc_generic.sh > gpurun_out/r02_pmc_generic.txt'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmcg_$C -- \
      python benchmarks/generic_bench.py --steps 1 --warmup 1 > $OUT/pmcg_$C.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
def load(c):
    f = glob.glob("gpurun_out/pmcg_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
for k in sorted(set(F) | set(W)):
    if "gen_" not in k and "gemm" not in k and "row_" not in k:
        continue
    big = lambda v: [x for x in v if x > 20000] or v   # update launches (> 20 MB), not the rollout's small ones
    f, w = big(F.get(k, [0.0])), big(W.get(k, [0.0]))
    # coalesced streams: FETCH_SIZE counts half the bytes on gfx950 (profiles/r02_pmc_calibration.json)
    print(json.dumps({"kernel": k[-60:], "launches": len(f), "FETCH_SIZE_KB_mean": round(sum(f) / len(f), 1),
                      "WRITE_SIZE_KB_mean": round(sum(w) / len(w), 1),
                      "hbm_MB_per_launch": round((2 * sum(f) / len(f) + sum(w) / len(w)) / 1024, 1)}))
PY
rm -rf $OUT/pmcg_FETCH_SIZE $OUT/pmcg_WRITE_SIZE

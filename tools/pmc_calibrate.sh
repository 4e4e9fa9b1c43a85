#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE per byte by access pattern (tools/pmc_calibrate.hip): bash tools/pmc_calibrate.sh  (via gpurun)
# Two separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass); writes
# gpurun_out/r02_pmc_calibration.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/pmc_cal tools/pmc_calibrate.hip || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/cal_$C
  timeout 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/cal_$C -- /tmp/pmc_cal > gpurun_out/cal_$C.log 2>&1
done
python - <<'PY'
import csv, glob, json
from collections import defaultdict
out = {"bytes_per_launch": 512 << 20, "note": "counter value / algorithmic bytes of the launch, per kernel of tools/pmc_calibrate.hip"}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/cal_%s/**/*counter_collection.csv" % C, recursive=True)
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == C:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[C] = sum(v) / len(v)
        out[k][C + "_per_byte"] = sum(v) / len(v) / float(512 << 20)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r02_pmc_calibration.json", "w"), indent=1)
PY
rm -rf gpurun_out/cal_FETCH_SIZE gpurun_out/cal_WRITE_SIZE

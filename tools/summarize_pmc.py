"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) into per-kernel HBM traffic.

    python tools/summarize_pmc.py fetch.csv write.csv > profiles/rNN_pmc_hbm.json

Units: the counters are KB per dispatch.  gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B
per 128-B request for wide coalesced streams, i.e. it under-reports such streams by 2x - calibrated here on the
GAE scan whose read set is known exactly (3 arrays of T*L floats + the value bootstrap).  Both the raw and the x2
figures are reported for the tower kernels; `bench.py` quotes the raw sum as `roofline.traffic`."""
import csv
import json
import sys
from collections import defaultdict

# bench.py's workload (configs[1]): rows per launch and bytes per record, for the algorithmic figure
ROWS, REC_BYTES = 4096 * 128, 64
KEYS = [("gae_scan", "gae_scan_kernel"), ("adv_normalize_pack", "adv_normalize_pack_kernel"),
        ("ppo_tower_pair", "ppo_tower_pair_kernel"),  # both towers in one launch (the default build)
        ("ppo_tower_policy", "ppo_tower_kernel<1,"), ("ppo_tower_critic", "ppo_tower_kernel<0,"),
        ("rollout_fused", "rollout_kernel"), ("ppo_apply", "ppo_apply_kernel"), ("ppo_reduce_pair", "ppo_reduce_pair_kernel")]


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for key, pat in KEYS:
                if pat in name:
                    acc[key].append(float(row["Counter_Value"]))
                    break
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key, _ in KEYS:
        if key in fetch or key in write:
            f, w = fetch.get(key, []), write.get(key, [])
            out[key] = {"FETCH_SIZE_KB_mean": round(sum(f) / max(len(f), 1), 1), "launches_fetch": len(f),
                        "WRITE_SIZE_KB_mean": round(sum(w) / max(len(w), 1), 1), "launches_write": len(w)}
    towers = ("ppo_tower_pair",) if "ppo_tower_pair" in out else ("ppo_tower_policy", "ppo_tower_critic")
    if all(k in out for k in towers):
        raw = sum(out[k]["FETCH_SIZE_KB_mean"] + out[k]["WRITE_SIZE_KB_mean"] for k in towers)
        x2 = sum(2 * out[k]["FETCH_SIZE_KB_mean"] + out[k]["WRITE_SIZE_KB_mean"] for k in towers)
        out["orl_ppo_fwd_bwd_pair"] = {
            "hbm_bytes_per_launch_raw": int(raw * 1024), "hbm_bytes_per_launch_fetch_x2": int(x2 * 1024),
            "algorithmic_bytes_per_launch": 2 * ROWS * REC_BYTES + 2 * ROWS * 8 + 256 * 4 * (4626 + 4561),
            "note": "FETCH_SIZE/WRITE_SIZE from separate rocprofv3 --pmc passes (KB). gfx950 FETCH_SIZE counts 64 B per "
                    "128-B request for wide coalesced streams (exact x2 on the GAE stream); the tower kernels gather "
                    "64-B records by 16-B-per-lane DMA, uncalibrated, so both raw and x2 are given. Algorithmic = "
                    "records + int64 indices read by both towers + the 256 per-workgroup partial rows written "
                    "(4626 / 4561 floats per row for the policy / critic tower of configs[1])."}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) into per-kernel HBM traffic.

    python tools/summarize_pmc.py fetch.csv write.csv > profiles/rNN_pmc_hbm.json

Units: the counters are KiB per dispatch.  Calibration on gfx950 (tools/pmc_calibrate.hip, 512 MiB per pattern,
profiles/r02_pmc_calibration.json): WRITE_SIZE is exact; FETCH_SIZE reports exactly 1/2 of a COALESCED stream at 16 B,
4 B per lane and for the 16-B global_load_lds DMA alike (128-B requests tallied at 64 B, MI355X_MICROARCH.md), and is
EXACT for 64-byte rows gathered at scattered positions (one 64-B request per row).  Hence per kernel:
  streaming kernels (gae_scan, adv_normalize_pack, ppo_reduce_pair, rollout): bytes = 2 * FETCH + WRITE
  tower kernels: records are 64-B rows at permuted positions (exact), the int64 index stream is coalesced (1/2):
                 bytes = FETCH + (index bytes) / 2 + WRITE."""
import csv
import json
import sys
from collections import defaultdict

# bench.py's workload (configs[1]): rows per launch and bytes per record, for the algorithmic figure
ROWS, REC_BYTES = 4096 * 128, 64
KEYS = [("gae_scan", "gae_scan_kernel"), ("adv_normalize_pack", "adv_normalize_pack_kernel"),
        ("ppo_tower_pair", "ppo_tower_pair_kernel"),  # both towers in one launch (the default build)
        ("ppo_tower_policy", "ppo_tower_kernel<1,"), ("ppo_tower_critic", "ppo_tower_kernel<0,"),
        ("rollout_chain", "rollout2_kernel"), ("rollout_fused", "rollout_kernel"), ("ppo_apply", "ppo_apply_kernel"), ("ppo_reduce_pair", "ppo_reduce_pair_kernel")]


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
    for key in ("gae_scan", "adv_normalize_pack", "ppo_reduce_pair", "rollout_fused"):
        if key in out:  # coalesced streams: FETCH_SIZE is half the bytes (calibration above)
            out[key]["hbm_bytes_per_launch"] = int((2 * out[key]["FETCH_SIZE_KB_mean"] + out[key]["WRITE_SIZE_KB_mean"]) * 1024)
    towers = ("ppo_tower_pair",) if "ppo_tower_pair" in out else ("ppo_tower_policy", "ppo_tower_critic")
    if all(k in out for k in towers):
        raw = sum(out[k]["FETCH_SIZE_KB_mean"] + out[k]["WRITE_SIZE_KB_mean"] for k in towers)
        idx_bytes = 2 * ROWS * 8  # both towers stream the int64 permutation
        out["orl_ppo_fwd_bwd_pair"] = {
            "hbm_bytes_per_launch": int(raw * 1024) + idx_bytes // 2, "hbm_bytes_per_launch_raw": int(raw * 1024),
            "algorithmic_bytes_per_launch": 2 * ROWS * REC_BYTES + idx_bytes + 128 * 4 * (4626 + 4561),
            "note": "FETCH_SIZE / WRITE_SIZE (KiB) from separate rocprofv3 --pmc passes. Calibrated (tools/pmc_calibrate.hip, "
                    "profiles/r02_pmc_calibration.json): WRITE_SIZE exact; FETCH_SIZE exact for the 64-B records gathered "
                    "at permuted rows, 1/2 for the coalesced int64 index stream - hbm_bytes_per_launch adds the missing "
                    "half of the index bytes to the raw sum. Algorithmic = records + indices read by both towers + the "
                    "2 x 128 per-workgroup partial rows written (4626 / 4561 floats per row for the policy / critic tower of "
                    "configs[1]); the second tower finds part of the shared record lines in L2 / Infinity Cache."}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env bash
# Round 5, GPU call 1 (gpurun --timeout 1500 -- 'bash tools/r05_calls/r05_call1.sh'): dot2c split probe, same-box A/B of the tower
# variants, LDS bank-conflict counters per image stride, parity tests of the touched kernels.
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. probe
hipcc --offload-arch=gfx950 -O3 tools/split_dot2c_probe.hip -o /tmp/split_probe > /dev/null 2>&1 && /tmp/split_probe > $OUT/split_dot2c_probe.txt 2>&1
cat $OUT/split_dot2c_probe.txt
# 2. A/B of the headline bench (two alternations)
: > $OUT/ab_tower.txt
for rep in 1 2; do
  for v in r04like dot2c_only new72 new80; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'], 'frac', r['roofline']['frac'])" >> $OUT/ab_tower.txt
  done
done
# 512-env shard + the wide-observation shapes, old vs new
for v in r04like new80; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'envs512 ms_per_step', r['ms_per_step'], 'pair_ms', r['roofline']['launch_ms'])" >> $OUT/ab_tower.txt
  python benchmarks/shape_sweep.py --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: continue
    print('$v', r.get('bench'), 'pair_ms', r.get('tower_pair_ms'), 'frac', r.get('tower_pair_frac_of_fp32_mfma_peak'), 'iter_ms', r.get('ms_per_iteration'))" >> $OUT/ab_tower.txt
done
cat $OUT/ab_tower.txt
# 3. LDS counters per stride
for v in new72 new80; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_$v -- \
     python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_$v.log 2>&1
  f=$(find $OUT/pmc_$v -name '*counter_collection.csv' | head -1)
  python - "$f" "$v" <<'PY' | tee -a $OUT/pmc_lds.txt
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ppo_tower_pair_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(sys.argv[2], k, round(sum(v)/len(v),1), len(v))
PY
  rm -rf $OUT/pmc_$v
done
# 4. parity tests on the default build
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ppo_update_gpu.py tests/test_gen_tower_gpu.py tests/test_kernels_gpu.py tests/test_a2c.py -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
timeout 300 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "bench" 2>&1 | tail -5 >> $OUT/pytest_subset.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $OUT/pytest_subset.log
timeout 300 python benchmarks/other_configs.py --steps 2 --warmup 1 2>/dev/null > $OUT/other_configs.jsonl
cut -c1-400 $OUT/other_configs.jsonl

#!/usr/bin/env bash
# round 6, call 1: the LayerNorm guard - adversarial tests on the guarded (default) and the unguarded (variants/ln_noguard.so) build,
# and what the guard costs on the headline / 512-env lines (alternated)
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['launch_ms'])"; }
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_layernorm_adversarial_gpu.py -m gpu -q -x 2>&1 | tail -15 > $OUT/r06_ln_guarded.log
cat $OUT/r06_ln_guarded.log
cp variants/ln_noguard.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_layernorm_adversarial_gpu.py -m gpu -q 2>&1 | tail -40 > $OUT/r06_ln_unguarded.log
tail -30 $OUT/r06_ln_unguarded.log | cut -c1-220
for i in 1 2 3; do
  for v in default ln_noguard; do
    cp variants/$v.so openrl_amd/csrc/liborl_hip.so
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | line "$v 4096"
    timeout 300 python bench.py --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | line "$v 512"
  done
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

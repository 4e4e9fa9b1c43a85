#!/usr/bin/env bash
# round 6, call 24: two ranks on ONE GPU (gloo rendezvous, IPC exchange) - the fp16 build against the bf16 x 3 build, alternating
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default bf16x3 default bf16x3; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; ORL_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['value'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so

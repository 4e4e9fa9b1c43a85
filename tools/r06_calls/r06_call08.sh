#!/usr/bin/env bash
# round 6, call 8: phase profiles out of the timing build of the final tree (tower pair, recurrent row kernel, chain rollout)
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
python tools/tower_phase_prof.py 2>/dev/null > $OUT/r06_tower_phase_prof.txt
python tools/tower_phase_prof.py --obs 18 --act 9 --T 200 2>/dev/null >> $OUT/r06_tower_phase_prof.txt
python tools/rnn_phase_prof.py fp32 2>/dev/null | grep -v "^{" > $OUT/r06_rnn_phase_prof.txt
python tools/rollout2_phase_prof.py 2>/dev/null | grep -v "^{" > $OUT/r06_rollout2_phase_prof.txt
python tools/rollout2_phase_prof.py --env cartpole 2>/dev/null | grep -v "^{" >> $OUT/r06_rollout2_phase_prof.txt
python tools/rollout2_phase_prof.py --envs 512 2>/dev/null | grep -v "^{" >> $OUT/r06_rollout2_phase_prof.txt
head -16 $OUT/r06_tower_phase_prof.txt; cat $OUT/r06_rollout2_phase_prof.txt | head -14
cp variants/default.so openrl_amd/csrc/liborl_hip.so

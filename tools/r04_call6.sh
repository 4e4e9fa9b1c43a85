#!/usr/bin/env bash
# round-4 GPU call 6: recurrent row kernel variants (parity, time, phase profile, PMC), general-tower no-spill build, torch launches
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp variants/r4d.so openrl_amd/csrc/liborl_hip.so
timeout 600 python -m pytest tests/test_rnn_train_gpu.py -q -x 2>&1 | tail -4
for g in split fp32 split_w4 split fp32 split_w4; do
  timeout 120 python benchmarks/rnn_update_bench.py --tower-gemm $g 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rnn_update', '$g', round(r['ms_per_epoch'],4))"
done
python tools/find_torch_launches.py 2>&1 | tail -25 > gpurun_out/r04_torch_launches.txt; cat gpurun_out/r04_torch_launches.txt
for v in r4d gt4; do cp variants/$v.so openrl_amd/csrc/liborl_hip.so; python benchmarks/generic_bench.py --steps 5 --warmup 3 2>/dev/null | grep generic_tower_path | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', {k: r[k] for k in r if 'ms' in k})"; done
cp variants/prof3.so openrl_amd/csrc/liborl_hip.so
(for g in split fp32 split_w4; do python tools/rnn_phase_prof.py $g 2>&1 | grep -v "^{" ; done) > gpurun_out/r04_rnn_phase_prof.txt 2>&1; cat gpurun_out/r04_rnn_phase_prof.txt
cp variants/r4d.so openrl_amd/csrc/liborl_hip.so
(bash tools/pmc_rnn_row.sh split; bash tools/pmc_rnn_row.sh fp32; bash tools/pmc_rnn_row.sh split_w4) > gpurun_out/r04_pmc_rnn.txt 2>&1; cat gpurun_out/r04_pmc_rnn.txt | grep -v wgrad

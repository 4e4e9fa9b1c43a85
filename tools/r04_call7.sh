#!/usr/bin/env bash
# round-4 GPU call 7: flag-synchronised streamed row kernel + train_info without torch launches: full GPU suite, timings, profile
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp variants/r4e.so openrl_amd/csrc/liborl_hip.so
timeout 200 python -m pytest tests/test_rnn_train_gpu.py -q -x -k "full_size or recurrent_train_matches" 2>&1 | tail -4
for g in split fp32 split_w4 split fp32 split_w4; do
  timeout 120 python benchmarks/rnn_update_bench.py --tower-gemm $g 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rnn_update', '$g', round(r['ms_per_epoch'],4))"
done
for k in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench', r['ms_per_step'], r['roofline']['launch_ms'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench envs512', r['ms_per_step'], r['roofline']['launch_ms'])"
done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04c_pytest.log; tail -6 gpurun_out/r04c_pytest.log
cp variants/prof4.so openrl_amd/csrc/liborl_hip.so
(for g in split split_w4; do python tools/rnn_phase_prof.py $g 2>&1 | grep -v "^{" ; done; python tools/rollout_phase_prof.py 2>&1 | grep -v "^{") > gpurun_out/r04_phase_prof4.txt 2>&1; cat gpurun_out/r04_phase_prof4.txt

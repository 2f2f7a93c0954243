#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
LAS_PARITY_LOG=$PWD/gpurun_out/r5b_parity.jsonl timeout 1500 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_rnn_seq.py tests/test_gpu_timed_geometry.py tests/test_gpu_las_parity.py tests/test_gpu_full_scale.py tests/test_gpu_dp.py -q -rs -x -s > gpurun_out/r5b_pytest.log 2>&1; tail -5 gpurun_out/r5b_pytest.log
for i in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop 2> gpurun_out/r5b_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prepared', d['ms_per_step'], d.get('schedule'))"
LAS_TAIL_WINDOW=160 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop 2>> gpurun_out/r5b_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail windows following the sweep', d['ms_per_step'])"
done
LAS_PHASES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2> gpurun_out/r5b_phases.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -3
python3 tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('f32 ms_per_step', d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if 'phase' in k or 'speller' in k})"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 ms_per_step', d['ms_per_step'])"

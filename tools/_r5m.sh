#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_las_parity.py tests/test_gpu_full_scale.py tests/test_gpu_rnn_seq.py -q -x 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('f32 ms_per_step', d['ms_per_step'], d['schedule'], {k:v for k,v in d['kernel_ms'].items() if 'phase' in k or 'fwd' in k})"
LAS_XPROJ_CHUNK=0 python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nochunk f32 ms_per_step', d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if 'phase:listener' in k})"
done

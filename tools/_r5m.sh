#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_las_parity.py tests/test_gpu_full_scale.py -q -x 2>&1 | tail -2
python3 tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('f32 ms_per_step', d['ms_per_step'])"
done
rm -f /tmp/par.jsonl; LAS_PARITY_LOG=/tmp/par.jsonl python3 -m pytest tests/test_gpu_full_scale.py -q 2>&1 | tail -1
python3 - <<'PY'
import json
for l in open('/tmp/par.jsonl'):
    d=json.loads(l)
    if d.get('test')=='full_T_train_step' and d['prec']=='f32':
        print(d['prec'], d['cell'], 'grad %.6e logits %.6e alphas %.6e loss %.6e' % (d['worst_grad_err'], d['logits'], d['alphas'], d['loss']))
PY

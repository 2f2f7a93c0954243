#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -f /tmp/par.jsonl
LAS_PARITY_LOG=/tmp/par.jsonl python3 -m pytest tests/test_gpu_full_scale.py -q 2>&1 | tail -2
python3 - <<'PY'
import json
for l in open('/tmp/par.jsonl'):
    d=json.loads(l)
    if d.get('test')=='full_T_train_step':
        print(d['prec'], d['cell'], 'grad %.6e logits %.6e alphas %.6e loss %.6e' % (d['worst_grad_err'], d['logits'], d['alphas'], d['loss']))
PY

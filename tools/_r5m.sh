#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests/test_gpu_lm.py tests/test_gpu_decode_timed_mode.py tests/test_gpu_beam_loop.py -q -x 2>&1 | tail -3
python3 tools/bench_cell_rows.py 2>&1 | grep "M ="
for i in 1 2; do
python3 bench.py --decode-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); d=d.get('decode',d)
print('b16', d['value_b16'], 'b64', d['value_b64'], 'us/step', d['us_per_decode_step'], d['utterances_per_batch'])"
done

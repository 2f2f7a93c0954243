#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests/test_gpu_lm.py tests/test_gpu_decode_timed_mode.py tests/test_gpu_beam_loop.py -q 2>&1 | tail -3
for i in 1 2 3; do
for L in liblas_hip_base.so liblas_hip.so; do
LAS_LIB_PATH=$GRAFT_REPO_ROOT/automatic-speech-recognition_amd/lib/$L python3 bench.py --decode-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); d=d.get('decode',d)
print('$L', 'b16', d['value_b16'], 'b64', d['value_b64'], 'us/step', d['us_per_decode_step'], d['step_parts_us'])"
done; done

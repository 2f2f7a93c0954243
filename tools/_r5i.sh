#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_lm.py tests/test_gpu_decode_timed_mode.py tests/test_gpu_beam_loop.py tests/test_gpu_beam_attention.py tests/test_gpu_cli.py -q -rs -x > gpurun_out/r5i_pytest.log 2>&1; tail -3 gpurun_out/r5i_pytest.log
python3 bench.py --decode-only 2> gpurun_out/r5i_dec.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['decode']
print({k: d.get(k) for k in ('value','value_b16','value_b64','us_per_decode_step','step_parts_us','timing','value_timing','phases_s','ragged','utterances_per_batch')})"

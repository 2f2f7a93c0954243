#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_dropout.py tests/test_gpu_rnn_seq.py::test_a_reported_timeout_keeps_the_optimiser_from_using_the_step tests/test_gpu_residency.py tests/test_gpu_las_parity.py::test_dropout_changes_activations_only_in_training -q -rs 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-300
for i in 1 2 3; do
  for v in 0 1; do
    LAS_NO_XCD_LOCAL_ROWS=$v python3 bench.py --decode-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['decode']
print('noxcd=$v', d['value'], d['value_b16'], d['us_per_decode_step'], d['step_parts_us'])"
  done
done
for i in 1 2; do
  for v in 0 1; do
    LAS_NO_STEP_RECOVERY=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('norecovery=$v', d['ms_per_step'], d.get('parity_mode',{}).get('ms_per_step'))"
  done
done

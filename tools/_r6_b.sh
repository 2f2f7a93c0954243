#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
LAS_PARITY_LOG=$PWD/gpurun_out/r6b_parity.jsonl timeout 1500 python3 -m pytest tests/test_gpu_run_sh_recipe.py "tests/test_gpu_speller_bf16.py::test_bf16_row_kernels_match_oracle" tests/test_gpu_speller_bf16.py::test_location_aware_loop_kernels_match_oracle -q -rs -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r6b_pytest.log
tail -60 gpurun_out/r6b_pytest.log
timeout 600 python3 bench.py --only-leg run_sh --steps 5 --warmup 2 > gpurun_out/r6b_run_sh.json 2> gpurun_out/r6b_run_sh.err; cat gpurun_out/r6b_run_sh.json; tail -3 gpurun_out/r6b_run_sh.err
for c in rnn lstm; do
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_rs_$c -o b -- python3 bench.py --only-leg run_sh_$c --steps 5 --warmup 2 > gpurun_out/r6b_runsh_${c}_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_rs_$c 2 gpurun_out/r6b_runsh_${c}_kernel_stats.csv > /dev/null
head -32 gpurun_out/r6b_runsh_${c}_kernel_stats.csv | cut -c1-160
done

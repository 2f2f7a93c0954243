#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
LAS_PARITY_LOG=$PWD/gpurun_out/r6a_parity.jsonl timeout 1500 python3 -m pytest tests/test_gpu_run_sh_recipe.py tests/test_gpu_full_scale.py::test_config3_location_aware_full_T_full_U tests/test_gpu_timed_geometry.py::test_train_eval_train_uses_the_updated_recurrent_weights -q -rs -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r6a_pytest.log
tail -40 gpurun_out/r6a_pytest.log
timeout 600 python3 bench.py --only-leg run_sh --steps 5 --warmup 2 > gpurun_out/r6a_run_sh.json 2> gpurun_out/r6a_run_sh.err; cat gpurun_out/r6a_run_sh.json; tail -3 gpurun_out/r6a_run_sh.err
timeout 600 python3 bench.py --only-leg config2_sampling --steps 10 --warmup 3 > gpurun_out/r6a_sampling.json 2> gpurun_out/r6a_sampling.err; cat gpurun_out/r6a_sampling.json; tail -3 gpurun_out/r6a_sampling.err
for c in rnn lstm; do
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_rs_$c -o b -- python3 bench.py --only-leg run_sh_$c --steps 5 --warmup 2 > gpurun_out/r6a_runsh_${c}_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_rs_$c 2 gpurun_out/r6a_runsh_${c}_kernel_stats.csv > /dev/null
head -25 gpurun_out/r6a_runsh_${c}_kernel_stats.csv | cut -c1-200
done

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
LAS_PARITY_LOG=$PWD/gpurun_out/r6c_parity.jsonl timeout 2400 python3 -m pytest tests -m gpu -q -rs -x 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c_pytest.log
tail -40 gpurun_out/r6c_pytest.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err; tail -c 1500 gpurun_out/r6c_bench.json; tail -3 gpurun_out/r6c_bench.err

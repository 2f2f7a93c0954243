#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=r5
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_f32_$P -o b -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-train-loop > $GRAFT_REPO_ROOT/gpurun_out/${P}_f32_kt.log 2>&1)
python3 tools/kernel_stats.py /tmp/kt_f32_$P 1 gpurun_out/${P}_f32_kernel_stats.csv > /dev/null
python3 tools/timeline.py "$(find /tmp/kt_f32_$P -name '*_results.db' | head -1)" --list > gpurun_out/${P}_f32_timeline.txt 2>&1
python3 tools/bench_gemm_f32.py > gpurun_out/${P}_f32_gemm_shapes.txt 2>&1
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_bench_f32.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${P}_bench.json 2> /dev/null
LAS_PARITY_LOG=$PWD/gpurun_out/${P}_parity_full_T.jsonl.new python3 -m pytest tests -m gpu -q -rs > gpurun_out/${P}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${P}_pytest_gpu.log
cut -c1-300 gpurun_out/${P}_bench_f32.json

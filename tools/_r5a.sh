#!/bin/bash
# round 5, first GPU pass: the suite, the bench line, one timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
LAS_PARITY_LOG=$PWD/gpurun_out/r5a_parity.jsonl timeout 1500 python3 -m pytest tests -m gpu -x -q -rs > gpurun_out/r5a_pytest.log 2>&1; tail -5 gpurun_out/r5a_pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; tail -c 400 gpurun_out/r5a_bench.json
LAS_PHASES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2> gpurun_out/r5a_phases.txt
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl_a -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > $GRAFT_REPO_ROOT/gpurun_out/r5a_tl.log 2>&1)
python3 tools/timeline.py "$(find /tmp/tl_a -name '*.db' | head -1)" --list > gpurun_out/r5a_timeline.txt 2>&1
head -30 gpurun_out/r5a_timeline.txt | cut -c1-200

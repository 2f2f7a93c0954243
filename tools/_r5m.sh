#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_f32 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2>&1)
python3 tools/timeline.py "$(find /tmp/kt_f32 -name '*_results.db' | head -1)" --list > gpurun_out/r5_f32_timeline.txt 2>&1

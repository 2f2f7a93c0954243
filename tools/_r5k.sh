#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_speller_bf16.py tests/test_gpu_configs.py -q -x 2>&1 | tail -2
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_k -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2>&1)
python3 tools/kernel_stats.py /tmp/kt_k 3 /tmp/ks.csv > /dev/null; grep -i "dkeys\|ce_rows" /tmp/ks.csv | cut -c1-120
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_k3 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2>&1)
python3 tools/kernel_stats.py /tmp/kt_k3 3 /tmp/ks3.csv > /dev/null; grep -i "dkeys\|ce_rows" /tmp/ks3.csv | cut -c1-120

#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1; do
(cd /tmp && LAS_NO_PREPARED_SWEEPS=$v rocprofv3 --kernel-trace -d /tmp/tl_e$v -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > $GRAFT_REPO_ROOT/gpurun_out/r5e_tl$v.log 2>&1)
python3 tools/timeline.py "$(find /tmp/tl_e$v -name '*.db' | head -1)" --list > gpurun_out/r5e_timeline_noprep$v.txt 2>&1
done
grep -n "rnn_seq\|pack_whh\|seq_prepare\|step window" gpurun_out/r5e_timeline_noprep0.txt | cut -c1-120 | head -30
echo ====
grep -n "rnn_seq\|pack_whh\|seq_prepare\|step window" gpurun_out/r5e_timeline_noprep1.txt | cut -c1-120 | head -30

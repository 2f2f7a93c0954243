#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in a b c; do
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl_f$v -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > /tmp/tl_f$v.log 2>&1)
python3 tools/timeline.py "$(find /tmp/tl_f$v -name '*.db' | head -1)" --list > /tmp/tl_f$v.txt 2>&1
grep -n "step window\|rnn_seq_fwd_hw" /tmp/tl_f$v.txt | head -3 | cut -c1-100
grep -n "idle between" /tmp/tl_f$v.txt | head -1 | cut -c1-160
done

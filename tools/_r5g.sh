#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp
NUTT=64 rocprofv3 --kernel-trace --stats -d /tmp/kt_dec64 -o b -- python3 $GRAFT_REPO_ROOT/tools/probe_decode_step.py > $GRAFT_REPO_ROOT/gpurun_out/r5g_kt64.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r5g_kt64.log | cut -c1-300
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/kt_dec64/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, queue_id, start, end from kernels order by start"))
out = 0
for i, r in enumerate(rows):
    if "dec_step_fwd_pf_kernel" in r[0] and i + 6 < len(rows) and out < 2 and i > len(rows) // 2:
        t0 = r[2]; print("---")
        for j in range(6):
            q = rows[i + j]; print("%8.1f us  %7.1f us  %s" % ((q[2] - t0) / 1e3, (q[3] - q[2]) / 1e3, q[0][:70]))
        out += 1
PY

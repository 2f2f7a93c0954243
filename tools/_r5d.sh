#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt_dec3 -o b -- python3 $GRAFT_REPO_ROOT/tools/probe_decode_step.py > $GRAFT_REPO_ROOT/gpurun_out/r5d_kt3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py "$(find /tmp/kt_dec3 -name '*_results.db' | head -1)" gpurun_out/r5d_decode_kernel_stats.csv > /dev/null 2>&1
head -14 gpurun_out/r5d_decode_kernel_stats.csv | cut -c1-160
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/kt_dec3/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, queue_id, start, end from kernels order by start"))
# find a run of consecutive (pf_lm, pair, beam_loop) triples and of 5-launch sequences; print a few with gaps
def show(pred_first, n):
    out = 0
    for i, r in enumerate(rows):
        if pred_first(r[0]) and i + n < len(rows) and out < 3 and i > 20000:
            t0 = r[2]
            print("---")
            for j in range(n + 1):
                q = rows[i + j]
                print("%8.1f us  %7.1f us  %s" % ((q[2] - t0) / 1e3, (q[3] - q[2]) / 1e3, q[0][:70]))
            out += 1
show(lambda n: "pf_lm_kernel" in n, 3)
show(lambda n: "dec_step_fwd_pf_kernel" in n, 5)
PY

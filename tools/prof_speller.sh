#!/bin/bash
# rocprofv3 kernel trace of tools/bench_speller.py; prints the top kernels.  Run on the GPU box: bash tools/prof_speller.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sp -o sp -- python3 $R/tools/bench_speller.py --iters 5 "$@" > $R/gpurun_out/prof_sp.log 2>&1
python3 - <<PY
import sqlite3
con=sqlite3.connect('$R/gpurun_out/prof_sp/sp_results.db')
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
    print("%-80s %6d %9.1f us avg %7.2f %5.1f%%"%(r[0][:80],r[1],r[2],r[3],r[4]))
PY

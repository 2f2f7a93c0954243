#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary table kept under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_results.db profiles/r1_xxx.csv [steps]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
con = sqlite3.connect(db)
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "calls_per_step", "total_ms", "avg_us", "percent", "ms_per_step"])  # top_kernels durations are in us
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 120 else name[:117] + "..."
        w.writerow([short, calls, round(calls / steps, 2), round(total / 1e3, 2), round(avg, 2), round(pct, 2),
                    round(total / 1e3 / steps, 2)])
print("wrote", out, len(rows), "kernels")

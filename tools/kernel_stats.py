#!/usr/bin/env python
"""Per-step kernel statistics from a `rocprofv3 --kernel-trace` run of bench.py (results .db): per kernel the launches per
train step, microseconds per step, average duration -- over the timed steps only (after the warm-up steps, delimited by the
clip_adam_kernel that ends every step).   python tools/kernel_stats.py <dir-with-results.db> <warmup> profiles/r2_kernel_stats.csv"""
import collections
import csv
import glob
import os
import sqlite3
import subprocess
import sys

d, warm, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
db = (glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True) + glob.glob(os.path.join(d, "*.db")))[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
idx = [i for i, r in enumerate(rows) if 'clip_adam' in r[0]]
lo, hi, n = idx[warm - 1] + 1, idx[-1] + 1, len(idx) - warm
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in rows[lo:hi]:
    agg[name][0] += 1
    agg[name][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
wall = (rows[hi - 1][2] - rows[lo][1]) / 1e3 / n


def dem(k):
    k = k.replace('.kd', '')
    try:
        r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', k], capture_output=True, text=True).stdout.strip()
        return r or k
    except Exception:
        return k


with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls_per_step', 'us_per_step', 'avg_us', 'percent_of_kernel_time'])
    w.writerow(['# %d timed steps; wall %.1f us/step; sum of kernel time %.1f us/step (side-stream work overlaps)' % (n, wall, tot / n), '', '', '', ''])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        w.writerow([dem(k)[:120], round(v[0] / n, 1), round(v[1] / n, 1), round(v[1] / v[0], 2), round(100 * v[1] / tot, 2)])
print(open(out).read()[:2500])

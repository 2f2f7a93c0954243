#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 kernel trace (rocpd sqlite): per-queue busy time by kernel
family and the idle gaps on the critical queue.   python tools/timeline.py <results.db> [--list]"""
import collections
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
rows = list(con.execute("select name, queue_id, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "clip_adam" in r[0]]
if len(adam) < 3:
    raise SystemExit("need at least 3 steps in the trace")
lo, hi = adam[-2] + 1, adam[-1] + 1          # kernels after the previous step's Adam up to this step's Adam
step = rows[lo:hi]
t0, t1 = step[0][2], step[-1][3]
print("step window: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))


def fam(n):
    for k in ("rnn_seq_fwd", "rnn_seq_bwd", "dec_loop_fwd", "dec_loop_bwd", "gemm_kk", "dec_step_fwd", "dec_step_bwd", "skinny_rows", "gemm_bf16_fast", "gemm_bf16_kernel",
              "gemm_f32", "splitk", "colsum", "pack_whh", "ce_rows", "clip_adam", "sumsq", "dkeys", "emb_grad", "to_bf16", "pair_rows",
              "tanh_bwd", "skinny_pack"):
        if k in n:
            return k
    return "other:" + n[:40]


byq = collections.defaultdict(list)
for n, q, s, e in step:
    byq[q].append((s, e, n))
for q, ks in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ks)
    print("\nqueue %s: %d kernels, busy %.3f ms" % (q, len(ks), busy / 1e6))
    agg = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in ks:
        agg[fam(n)] += e - s
        cnt[fam(n)] += 1
    for k, v in agg.most_common(14):
        print("   %-28s %8.3f ms  x%d" % (k, v / 1e6, cnt[k]))
    gaps = 0
    big = []
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        g = s1 - e0
        if g > 0:
            gaps += g
            if g > 20000:
                big.append((g, fam(n0), fam(n1)))
    print("   idle between its kernels: %.3f ms; gaps > 20 us:" % (gaps / 1e6), [(round(g / 1e3, 1), a, b) for g, a, b in big][:20])
if "--list" in sys.argv:
    for n, q, s, e in step:
        print("%9.1f us  q%s  %7.1f us  %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, n[:100]))

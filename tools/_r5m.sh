#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_rnn_seq.py -q -x 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('f32 ms_per_step', d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if 'rnn_seq_bwd' in k or 'phase:back' in k})"
done
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pmc_g -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm_f32.py > /tmp/pmc_g.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sqlite3, glob, collections
db = glob.glob('/tmp/pmc_g/**/*_results.db', recursive=True)
print(db)
con = sqlite3.connect(db[0])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if 'pmc_event' in t][0]; info=[t for t in tabs if 'info_pmc' in t][0]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name"
res = collections.defaultdict(dict)
for kn, cn, v, n in con.execute(q):
    if 'mf32' in kn: res[kn[:60]][cn] = v / max(n,1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()): print('   %-28s %14.0f' % (c, v))
PY

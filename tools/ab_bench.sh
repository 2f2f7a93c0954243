#!/bin/bash
# A/B of two environment settings on ONE box: tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [reps] [steps]   (alternating runs, ms per step of each)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
A="$1"; B="$2"; R=${3:-4}; S=${4:-100}
for i in $(seq 1 $R); do
  for v in "$A" "$B"; do
    ms=$(env $v python3 bench.py --steps $S --warmup 5 --no-cpu-baseline --no-decode --no-train-loop 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v $ms"
  done
done | python3 -c "
import sys, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    k, v = l.rsplit(' ', 1); d[k].append(float(v))
for k, v in d.items():
    v2 = sorted(v); print('%-40s median %.3f  all %s' % (k, v2[len(v2)//2], ' '.join('%.3f' % x for x in v)))
"

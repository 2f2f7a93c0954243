#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo early; ./tools/micro/bin/bench_fused 0 loc 2>&1 | grep "one launch"
echo noearly; ./tools/micro/bin/bench_fused_noearly 0 loc 2>&1 | grep "one launch"
done

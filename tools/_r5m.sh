#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for A in 0 1 2 4 6; do
  L=$GRAFT_REPO_ROOT/automatic-speech-recognition_amd/lib/liblas_hip_lb$A.so; [ $A = 0 ] && L=$GRAFT_REPO_ROOT/automatic-speech-recognition_amd/lib/liblas_hip.so
  echo "LB_ABL=$A"; LAS_LIB_PATH=$L python3 tools/bench_cell_rows.py 2>&1 | grep "M = 1024"
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for A in 0 1 2 3; do
  L=$GRAFT_REPO_ROOT/automatic-speech-recognition_amd/lib/liblas_hip_ablg$A.so; [ $A = 0 ] && L=$GRAFT_REPO_ROOT/automatic-speech-recognition_amd/lib/liblas_hip.so
  echo "ABL=$A"; LAS_LIB_PATH=$L python3 tools/bench_gemm_f32.py 2>&1 | grep "x-proj L1\|dX x-proj\|dW_ih\|dW_hh"
done

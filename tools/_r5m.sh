#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LAS_PARITY_LOG=$PWD/gpurun_out/r5_parity_full_T.jsonl python3 -m pytest tests -m gpu -q -rs > gpurun_out/r5_pytest_gpu.log 2>&1; tail -3 gpurun_out/r5_pytest_gpu.log

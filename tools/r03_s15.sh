#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03n
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -n 30 $OUT/pytest_gpu.log

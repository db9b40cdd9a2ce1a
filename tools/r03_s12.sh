#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03l
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 600 python tools/world_projection.py --world 8 > $OUT/world8.log 2>&1
tail -n 60 $OUT/world8.log
timeout 600 python tools/world_projection.py --world 4 --steps 20 > $OUT/world4.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_mosasrec.py tests/test_gpu_configs.py tests/test_mosasrec_golden.py tests/test_gpu_eval.py tests/test_gpu_misc.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest_sel.log 2>&1
tail -n 8 $OUT/pytest_sel.log

#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02f
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
for v in 2 3; do
  ( PXR_TOPK_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py -m gpu -q -k "topk or trainer or harness or evaluate" 2>&1 | tail -6 ) > $OUT/pytest_topk_v$v.log 2>&1
done
for v in 1 2 3; do
  PXR_TOPK_VARIANT=$v timeout 300 python tools/eval_bench.py 2>&1 | grep -v amdgpu > $OUT/eval_bench_v$v.log
done
( timeout 300 python -m pytest tests/test_gpu_vit.py -m gpu -q 2>&1 | tail -3 ) > $OUT/pytest_vit.log 2>&1
tail -n 4 $OUT/pytest_topk_v2.log $OUT/pytest_topk_v3.log $OUT/pytest_vit.log; cat $OUT/eval_bench_v*.log

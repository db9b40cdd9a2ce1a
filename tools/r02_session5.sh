#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02e
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_mosasrec.py tests/test_mosasrec_golden.py tests/test_gpu_gemm.py -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_vit.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "vit" 2>&1 | tail -30 ) > $OUT/pytest_vit_full.log 2>&1
for enc in clip-vit-base-patch32 clip-vit-base-patch16; do
  timeout 300 python tools/pixelnet_bench.py $enc 2>&1 | grep -v amdgpu >> $OUT/pixelnet_step.log
done
cat $OUT/pytest_vit.log $OUT/pytest_vit_full.log $OUT/pixelnet_step.log

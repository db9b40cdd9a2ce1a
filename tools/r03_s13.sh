#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03m
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_sasrec.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_adamw.py tests/test_gpu_vit.py tests/test_gpu_configs.py tests/test_gpu_mosasrec.py -x -q -m gpu 2>&1 | tail -6 ) > $OUT/pytest_sel.log 2>&1
tail -n 4 $OUT/pytest_sel.log
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-gemm-events --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), round(d['ms_per_step'],4))"
done

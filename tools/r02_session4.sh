#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02d
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_lazy_adamw.py tests/test_gpu_sasrec.py tests/test_gpu_fullsize.py tests/test_gpu_misc.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest_subset.log 2>&1
B="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --dw-mode grouped"
timeout 300 python bench.py $B > $OUT/bench_look_thin.json 2> $OUT/bench_look_thin.err
PXR_PREFETCH_BLOCKS=512 timeout 300 python bench.py $B > $OUT/bench_look_512.json 2> $OUT/bench_look_512.err
PXR_PREFETCH_BLOCKS=128 timeout 300 python bench.py $B > $OUT/bench_look_128.json 2> $OUT/bench_look_128.err
PXR_PREFETCH_PRIO=0 timeout 300 python bench.py $B > $OUT/bench_look_prio0.json 2> $OUT/bench_look_prio0.err
timeout 300 python bench.py $B --no-lookahead > $OUT/bench_nolook.json 2> $OUT/bench_nolook.err
timeout 600 python bench.py --dw-mode grouped > $OUT/bench_default_grouped.json 2> $OUT/bench_default_grouped.err
cat $OUT/pytest_subset.log

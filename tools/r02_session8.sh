#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02h
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py -m gpu -q -k "topk or trainer or harness or evaluate" 2>&1 | tail -12 ) > $OUT/pytest_topk_v4.log 2>&1
timeout 300 python tools/eval_bench.py 2>&1 | grep -v amdgpu > $OUT/eval_bench_v4.log
PXR_TOPK_VARIANT=2 timeout 300 python tools/eval_bench.py 2>&1 | grep -v amdgpu > $OUT/eval_bench_v2.log
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_eval
rm -rf $P && mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o eval -- python $REPO/tools/eval_bench.py > $P/stdout.log 2>&1
rm -f $P/*.db $P/*kernel_trace.csv
tail -n 8 $OUT/pytest_topk_v4.log; cat $OUT/eval_bench_v4.log $OUT/eval_bench_v2.log; grep -E "score_|topk|gemm_kernel" $P/eval_kernel_stats.csv | cut -c1-200

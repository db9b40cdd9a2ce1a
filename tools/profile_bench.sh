#!/bin/bash
# rocprofv3 kernel-trace of the default bench command (run on the GPU box): summary -> gpurun_out/prof_bench/
# usage: bash tools/profile_bench.sh [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-gemm-events "$@" > $OUT/bench_stdout.log 2>&1
rm -f $OUT/*.db
ls $OUT

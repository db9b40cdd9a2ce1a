#!/bin/bash
# Round-2 GPU session 2: tests + A/B of (look-ahead, XCD tile order, weight-gradient placement) + sweep + trace + PMC.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02b
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
B="--steps 100 --warmup 10 --no-cpu-baseline --no-extras"
timeout 300 python bench.py $B --no-lookahead > $OUT/bench_nolook.json 2> $OUT/bench_nolook.err
timeout 300 python bench.py $B --dw-mode grouped > $OUT/bench_grouped.json 2> $OUT/bench_grouped.err
PXR_GEMM_XCD=0 timeout 300 python bench.py $B --dw-mode grouped > $OUT/bench_grouped_xcd0.json 2> $OUT/bench_grouped_xcd0.err
PXR_GEMM_XCD=0 timeout 300 python bench.py $B > $OUT/bench_xcd0.json 2> $OUT/bench_xcd0.err
for x in 0 1; do
  PXR_GEMM_XCD=$x TILES=64,3264,12861,1281 timeout 300 python tools/gemm_sweep.py > $OUT/gemm_sweep_xcd$x.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_bench
rm -rf $P && mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps 300 --no-cpu-baseline --no-extras --no-gemm-events > $P/bench_stdout.log 2>&1
rm -f $P/*.db
python $REPO/tools/trace_union.py $(ls $P/*kernel_trace.csv | head -1) 360 $OUT/trace_union.json > /dev/null 2>&1
gzip -f $P/*kernel_trace.csv 2>/dev/null
Q=$OUT/pmc
rm -rf $Q && mkdir -p $Q
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $Q -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph --dw-mode grouped > $Q/${c}_stdout.log 2>&1
done
rm -f $Q/*.db
python $REPO/tools/pmc_summarise.py $Q/summary.json $(ls $Q/*counter_collection.csv) > /dev/null 2>&1
rm -f $Q/*kernel_trace.csv
gzip -f $Q/*counter_collection.csv 2>/dev/null
tail -n 6 $OUT/pytest_gpu.log

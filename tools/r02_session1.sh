#!/bin/bash
# Round-2 GPU session 1 (run on the GPU box through gpurun): tests, bench A/B of the weight-gradient placement, rocprof
# kernel trace, GEMM tile sweep, PMC traffic of the step's GEMMs.  Everything lands in gpurun_out/r02/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu_configs.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for mode in grouped fork_layer fork_half; do
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --dw-mode $mode > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
done
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-graph > $OUT/bench_eager.json 2> $OUT/bench_eager.err
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --table-update dense > $OUT/bench_dense.json 2> $OUT/bench_dense.err
# rocprof kernel trace of the default command
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_bench
rm -rf $P && mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps 300 --no-cpu-baseline --no-extras --no-gemm-events > $P/bench_stdout.log 2>&1
rm -f $P/*.db
python $REPO/tools/trace_union.py $(ls $P/*kernel_trace.csv | head -1) 360 $OUT/trace_union.json > /dev/null 2>&1
# keep the trace small: the stats + the union summary are what gets committed
gzip -f $P/*kernel_trace.csv 2>/dev/null
# GEMM tile sweep at the step's shapes
cd $REPO
TILES=64,642,3264,12864,64128,128,1281,12861,1282 timeout 300 python tools/gemm_sweep.py > $OUT/gemm_sweep.log 2>&1
# PMC: HBM traffic of the step's GEMM kernels (separate FETCH / WRITE passes, eager so that every kernel is a dispatch)
cd /tmp
Q=$OUT/pmc
rm -rf $Q && mkdir -p $Q
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $Q -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > $Q/${c}_stdout.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $Q -o MFMA -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > $Q/MFMA_stdout.log 2>&1
rm -f $Q/*.db
python $REPO/tools/pmc_summarise.py $Q/summary.json $(ls $Q/*counter_collection.csv) > /dev/null 2>&1
rm -f $Q/*kernel_trace.csv
gzip -f $Q/*counter_collection.csv 2>/dev/null
ls -la $OUT $P $Q
tail -5 $OUT/pytest_gpu.log $OUT/pytest_gpu_configs.log

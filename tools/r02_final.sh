#!/bin/bash
# Round-2 final measurement session (one MI355X): full GPU suite, default bench line (+ the same in f32 GEMM mode),
# rocprofv3 kernel trace of the bench command, PMC HBM-traffic passes, eval / PixelNet benches.  Output: gpurun_out/r02final
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02final
rm -rf $OUT && mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
PXR_GEMM_MODE=f32 timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_f32_mode.json 2> $OUT/bench_f32_mode.err
timeout 200 python tools/eval_bench.py 2>&1 | grep -v amdgpu > $OUT/eval_bench.log
timeout 300 python bench.py --model pixelnet --no-cpu-baseline > $OUT/bench_pixelnet_b32.json 2> $OUT/bench_pixelnet_b32.err
timeout 300 python bench.py --force-collectives --no-graph --no-cpu-baseline --no-extras > $OUT/bench_force_collectives.json 2> $OUT/bench_force_collectives.err
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof_bench
mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps 300 --no-cpu-baseline --no-extras --no-gemm-events > $P/bench_stdout.log 2>&1
rm -f $P/*.db
python $REPO/tools/trace_union.py $(ls $P/*kernel_trace.csv | head -1) 360 $OUT/trace_union.json > /dev/null 2>&1
rm -f $P/*kernel_trace.csv
Q=$OUT/pmc
mkdir -p $Q
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $Q -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > $Q/${c}_stdout.log 2>&1
done
rm -f $Q/*.db
python $REPO/tools/pmc_summarise.py $Q/summary.json $(ls $Q/*counter_collection.csv) > /dev/null 2>&1
python $REPO/tools/gemm_traffic.py $Q/summary.json $OUT/gemm_traffic_summary.json "final code, bf16x3 GEMM mode" > $OUT/gemm_traffic.log 2>&1
rm -f $Q/*kernel_trace.csv $Q/*counter_collection.csv
tail -n 4 $OUT/pytest_gpu.log
python - <<'PY'
import json
for f in ("bench_default", "bench_f32_mode", "bench_pixelnet_b32", "bench_force_collectives"):
    try:
        d = json.loads(open(f"/root/repo/gpurun_out/r02final/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), "gemm frac", round(r["frac"], 3), r.get("gemm_time_per_step_us"))
    except Exception as e:
        print(f, "failed", e)
PY
cat $OUT/eval_bench.log $OUT/gemm_traffic.log

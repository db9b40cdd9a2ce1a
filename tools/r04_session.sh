#!/bin/bash
# Round-4 measurement stages (one MI355X).  usage: bash tools/r04_session.sh <outdir-name> [stage ...]
#   gemmtest  tests/test_gpu_gemm_p3.py          vitgemm   tools/vit_gemm_bench.py + tools/big_gemm_tiles.py
#   pixel     PixelNet line                      pixelprof PixelNet rocprofv3 kernel stats
#   bench     default bench line                 prof      rocprofv3 kernel stats + timeline of the default step
#   tests     full GPU suite                     eval      full-sort eval bench (+ kernel stats)
#   b2048     rocprofv3 kernel stats at B = 2048
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$REPO/gpurun_out/$NAME
mkdir -p "$OUT"
STAGES=${*:-gemmtest vitgemm pixel}
export PYTHONUNBUFFERED=1
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd "$REPO"
if has gemmtest; then
  ( time timeout 900 python -m pytest tests/test_gpu_gemm_p3.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -15 ) > "$OUT/pytest_gemm_p3.log" 2>&1
  tail -n 8 "$OUT/pytest_gemm_p3.log"
fi
if has tests; then
  ( time timeout 2400 python -m pytest tests -m gpu -v 2>&1 | grep -v "amdgpu.ids\|socket.cpp" ) > "$OUT/pytest_gpu_full.log" 2>&1
  grep -v "PASSED\|^$" "$OUT/pytest_gpu_full.log" | cut -c1-400 | tail -n 60 > "$OUT/pytest_gpu.log"
  grep -c PASSED "$OUT/pytest_gpu_full.log"; grep "PASSED" "$OUT/pytest_gpu_full.log" | tail -2 | cut -c1-200; tail -n 40 "$OUT/pytest_gpu.log"
fi
if has vitgemm; then
  timeout 300 python tools/vit_gemm_bench.py > "$OUT/vit_gemm_bench.log" 2>&1; cat "$OUT/vit_gemm_bench.log"
  PXR_P4=0 timeout 300 python tools/vit_gemm_bench.py > "$OUT/vit_gemm_bench_p4off.log" 2>&1; cat "$OUT/vit_gemm_bench_p4off.log"
fi
if has pixel; then
  timeout 400 python bench.py --model pixelnet --no-cpu-baseline > "$OUT/bench_pixelnet_b16.json" 2> "$OUT/bench_pixelnet_b16.err"; cat "$OUT/bench_pixelnet_b16.json"; tail -3 "$OUT/bench_pixelnet_b16.err"
  PXR_P4=0 timeout 400 python bench.py --model pixelnet --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/bench_pixelnet_b16_p4off.json" 2> "$OUT/bench_pixelnet_b16_p4off.err"; cat "$OUT/bench_pixelnet_b16_p4off.json"
fi
if has bench; then
  timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cat "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
fi
if has eval; then
  timeout 400 python tools/eval_bench.py > "$OUT/eval_bench.log" 2>&1; tail -20 "$OUT/eval_bench.log"
fi
cd /tmp && export TMPDIR=/tmp
if has pixelprof; then
  P=$OUT/prof_pixelnet; mkdir -p "$P"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o pix -- python $REPO/bench.py --model pixelnet --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-gemm-events > "$P/stdout.log" 2>&1
  rm -f "$P/pix_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -25 "$P/pix_kernel_stats.csv" | cut -c1-200
fi
if has prof; then
  P=$OUT/prof_bench; mkdir -p "$P"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps 300 --no-cpu-baseline --no-extras --no-gemm-events > "$P/bench_stdout.log" 2>&1
  python $REPO/tools/step_timeline.py "$P/bench_kernel_trace.csv" "$OUT/step_timeline.txt" > "$P/timeline.log" 2>&1
  rm -f "$P/bench_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -40 "$P/bench_kernel_stats.csv" | cut -c1-200
fi
if has b2048; then
  P=$OUT/prof_b2048; mkdir -p "$P"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o b2048 -- python $REPO/bench.py --batch 2048 --steps 10 --warmup 3 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events > "$P/stdout.log" 2>&1
  rm -f "$P/b2048_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -40 "$P/b2048_kernel_stats.csv" | cut -c1-200
fi
cd "$REPO"
if has evaltest; then
  ( time timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_eval.py tests/test_gpu_misc.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -15 ) > "$OUT/pytest_eval.log" 2>&1
  tail -n 8 "$OUT/pytest_eval.log"
  timeout 300 python tools/eval_bench.py > "$OUT/eval_bench.log" 2>&1; grep -v amdgpu.ids "$OUT/eval_bench.log" | tail
  PXR_SCORE_P4=0 timeout 300 python tools/eval_bench.py > "$OUT/eval_bench_p4off.log" 2>&1; grep -v amdgpu.ids "$OUT/eval_bench_p4off.log" | tail
fi
cd /tmp && export TMPDIR=/tmp
if has pixelpmc; then
  Q=$OUT/pmc_pixelnet; mkdir -p "$Q"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python $REPO/bench.py --model pixelnet --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-gemm-events > "$Q/${c}_stdout.log" 2>&1
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$Q/summary.json" $(ls "$Q"/*counter_collection.csv) > /dev/null 2>&1
  python $REPO/tools/gemm_traffic.py "$Q/summary.json" "$OUT/pixelnet_gemm_traffic_summary.json" "round 4, PixelNet ViT-B/16 B=16 step, ping-pong tiles" > "$OUT/pixelnet_gemm_traffic.log" 2>&1
  rm -f "$Q"/*kernel_trace.csv "$Q"/*counter_collection.csv
  cat "$OUT/pixelnet_gemm_traffic.log"
fi
if has pmc64; then
  Q=$OUT/pmc64; mkdir -p "$Q"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > "$Q/${c}_stdout.log" 2>&1
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$Q/summary.json" $(ls "$Q"/*counter_collection.csv) > /dev/null 2>&1
  python $REPO/tools/gemm_traffic.py "$Q/summary.json" "$OUT/gemm_traffic_summary.json" "round 4 code, planes GEMM mode, B=64" > "$OUT/gemm_traffic.log" 2>&1
  rm -f "$Q"/*kernel_trace.csv "$Q"/*counter_collection.csv
  cat "$OUT/gemm_traffic.log"
fi

#!/bin/bash
# Round-3 measurement session (one MI355X).  usage: bash tools/r03_final.sh [stage ...]   (default: all stages)
#   tests   full GPU suite                                   bench   default bench line (+ f32 / bf16x3-in-loop GEMM modes)
#   prof    rocprofv3 kernel stats + one-step timeline       pmc     HBM traffic of the step's GEMM family (FETCH/WRITE passes)
#   sq      SQ / LDS / MFMA counters of the planes GEMMs      pixel   PixelNet line + kernel stats + HBM traffic
#   eval    full-sort eval bench + kernel stats               w8      one-GPU projection of the 8-rank step
# Output: gpurun_out/r03final (copy what is to be judged into profiles/r03).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03final
mkdir -p "$OUT"
STAGES=${*:-tests bench prof pmc sq pixel eval w8}
export PYTHONUNBUFFERED=1
has() { [[ " $STAGES " == *" $1 "* ]]; }
cd "$REPO"
if has tests; then
  ( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -12 ) > "$OUT/pytest_gpu.log" 2>&1
  tail -n 6 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  PXR_GEMM_MODE=f32 timeout 300 python bench.py --no-cpu-baseline --no-extras > "$OUT/bench_f32_mode.json" 2> "$OUT/bench_f32_mode.err"
  PXR_PLANES=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > "$OUT/bench_b3_inloop_mode.json" 2> "$OUT/bench_b3_inloop_mode.err"
  PXR_LAZY_REPLAY=exact timeout 300 python bench.py --no-cpu-baseline --no-extras > "$OUT/bench_exact_replay.json" 2> "$OUT/bench_exact_replay.err"
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  P=$OUT/prof_bench; mkdir -p "$P"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps 300 --no-cpu-baseline --no-extras --no-gemm-events > "$P/bench_stdout.log" 2>&1
  python $REPO/tools/step_timeline.py "$P/bench_kernel_trace.csv" "$OUT/step_timeline.txt" > "$P/timeline.log" 2>&1
  rm -f "$P/bench_kernel_trace.csv"; find "$P" -name "*.db" -delete
fi
if has pmc; then
  Q=$OUT/pmc; mkdir -p "$Q"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > "$Q/${c}_stdout.log" 2>&1
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$Q/summary.json" $(ls "$Q"/*counter_collection.csv) > /dev/null 2>&1
  python $REPO/tools/gemm_traffic.py "$Q/summary.json" "$OUT/gemm_traffic_summary.json" "round 3 final code, planes GEMM mode" > "$OUT/gemm_traffic.log" 2>&1
  rm -f "$Q"/*kernel_trace.csv "$Q"/*counter_collection.csv
  cat "$OUT/gemm_traffic.log"
fi
if has sq; then
  bash $REPO/tools/pmc_p3.sh r03final/pmc_p3 0 > "$OUT/pmc_p3.log" 2>&1
  rm -f "$OUT"/pmc_p3/*kernel_trace.csv "$OUT"/pmc_p3/*counter_collection.csv
  tail -n 5 "$OUT/pmc_p3.log"
fi
if has pixel; then
  cd "$REPO"
  timeout 400 python bench.py --model pixelnet --no-cpu-baseline > "$OUT/bench_pixelnet_b16.json" 2> "$OUT/bench_pixelnet_b16.err"
  cd /tmp
  P=$OUT/prof_pixelnet; mkdir -p "$P"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o pix -- python $REPO/bench.py --model pixelnet --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-gemm-events > "$P/stdout.log" 2>&1
  rm -f "$P/pix_kernel_trace.csv"; find "$P" -name "*.db" -delete
  Q=$OUT/pmc_pixelnet; mkdir -p "$Q"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python $REPO/bench.py --model pixelnet --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-gemm-events > "$Q/${c}_stdout.log" 2>&1
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$Q/summary.json" $(ls "$Q"/*counter_collection.csv) > /dev/null 2>&1
  python $REPO/tools/gemm_traffic.py "$Q/summary.json" "$OUT/pixelnet_gemm_traffic_summary.json" "round 3 final code, PixelNet ViT-B/16 B=16 step" > "$OUT/pixelnet_gemm_traffic.log" 2>&1
  rm -f "$Q"/*kernel_trace.csv "$Q"/*counter_collection.csv
  cat "$OUT/pixelnet_gemm_traffic.log"
fi
if has eval; then
  cd "$REPO"
  timeout 300 python tools/eval_bench.py 2>&1 | grep -v amdgpu > "$OUT/eval_bench.log"
  cd /tmp
  P=$OUT/prof_eval; mkdir -p "$P"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o eval -- python $REPO/tools/eval_bench.py > "$P/stdout.log" 2>&1
  rm -f "$P/eval_kernel_trace.csv"; find "$P" -name "*.db" -delete
  cat "$OUT/eval_bench.log"
fi
if has w8; then
  cd "$REPO"
  timeout 600 python tools/world_projection.py --world 8 > "$OUT/world_projection.log" 2>&1
  cp gpurun_out/world_projection_w8.json "$OUT/" 2>/dev/null
  tail -n 3 "$OUT/world_projection.log"
fi
python - <<PY
import json
for f in ("bench_default", "bench_f32_mode", "bench_b3_inloop_mode", "bench_exact_replay", "bench_pixelnet_b16"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), "gemm frac", round(r["frac"], 3), r.get("gemm_time_per_step_us"))
    except Exception as e:
        print(f, "not available:", e)
PY

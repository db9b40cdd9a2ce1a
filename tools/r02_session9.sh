#!/bin/bash
# full GPU suite + default bench with the bf16x3 GEMM mode, and the same bench in f32 mode (A/B)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02j
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_b3.json 2> $OUT/bench_b3.err
PXR_GEMM_MODE=f32 timeout 400 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_f32.json 2> $OUT/bench_f32.err
tail -n 25 $OUT/pytest_gpu.log
python - <<'PY'
import json
for f in ("bench_b3", "bench_f32"):
    try:
        d = json.loads(open(f"/root/repo/gpurun_out/r02j/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["value"]), round(d["ms_per_step"], 4), "gemm us/step", round(r["gemm_time_per_step_us"], 1), "frac", round(r["frac"], 3), "launches", r["launches_per_step"])
        for k, v in r["kernels"].items():
            print("   ", k, {a: round(b, 2) for a, b in v.items()})
    except Exception as e:
        print(f, "failed", e)
PY
tail -n 5 $OUT/bench_b3.err

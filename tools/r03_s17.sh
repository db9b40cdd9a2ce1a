#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03p
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_configs.py tests/test_gpu_mosasrec.py tests/test_mosasrec_golden.py -x -q -m gpu 2>&1 | tail -6 ) > $OUT/pytest_sel.log 2>&1
tail -n 5 $OUT/pytest_sel.log
timeout 600 python bench.py --model pixelnet --no-cpu-baseline > $OUT/bench_pixelnet_planes.json 2> $OUT/bench_pixelnet_planes.err
PXR_PLANES=0 timeout 600 python bench.py --model pixelnet --no-cpu-baseline > $OUT/bench_pixelnet_noplanes.json 2> $OUT/bench_pixelnet_noplanes.err
python - <<'PY'
import json
for f in ("planes", "noplanes"):
    try:
        d = json.loads(open(f"/root/repo/gpurun_out/r03p/bench_pixelnet_{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms/step", round(d["ms_per_step"], 2), "images/s", round(d["images_per_s"]), "gemm ms", round(r["gemm_time_per_step_ms"], 2), "frac", round(r["frac"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
tail -n 3 $OUT/bench_pixelnet_planes.err

#!/bin/bash
# Round-2 GPU session 3: dual-accumulator GEMM A/B (sweep + in-step) and SQ stall counters of the 64x64 GEMM.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02c
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
timeout 120 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -3 > $OUT/pytest_gemm.log
TILES=64,641,12861,128611 timeout 300 python tools/gemm_sweep.py > $OUT/gemm_sweep.log 2>&1
B="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --dw-mode grouped"
timeout 300 python bench.py $B > $OUT/bench_t64.json 2> $OUT/bench_t64.err
PXR_GEMM_SMALL_TILE=641 timeout 300 python bench.py $B > $OUT/bench_t641.json 2> $OUT/bench_t641.err
PXR_GEMM_SMALL_TILE=12861 timeout 300 python bench.py $B > $OUT/bench_t12861.json 2> $OUT/bench_t12861.err
PXR_GEMM_SMALL_TILE=128611 timeout 300 python bench.py $B > $OUT/bench_t128611.json 2> $OUT/bench_t128611.err
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gemm_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
dev = "cuda"
for tile in (64, 641):
    for (n, k) in ((1536, 512), (512, 512), (512, 1024)):
        x = torch.randn(3200, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
        y = torch.empty(3200, n, device=dev)
        for _ in range(4): ops.gemm(True, True, 3200, n, k, x, k, W, k, y, n, ops.EPI_BIAS, bias=b, use_ws=False, tile_hint=tile)
torch.cuda.synchronize()
PY
Q=$OUT/pmc
rm -rf $Q && mkdir -p $Q
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $Q -o a -- python /tmp/gemm_one.py > $Q/stdout_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $Q -o b -- python /tmp/gemm_one.py > $Q/stdout_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $Q -o c -- python /tmp/gemm_one.py > $Q/stdout_c.log 2>&1
rm -f $Q/*.db $Q/*kernel_trace.csv
python - <<'PY'
import csv, glob, json, os
from collections import defaultdict
Q = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r02c/pmc")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(Q + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" not in r["Kernel_Name"]:
            continue
        key = r["Kernel_Name"].split("(")[0].replace("void pxr::", "") + " grid=" + r["Grid_Size"]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(Q + "/sq_summary.json", "w"), indent=1, sort_keys=True)
PY
gzip -f $Q/*counter_collection.csv 2>/dev/null
cat $OUT/pytest_gemm.log; grep -v amdgpu $OUT/gemm_sweep.log | cut -c1-180

#!/bin/bash
# SQ / MFMA / LDS / L2 counters of the big-tile planes GEMMs: lockstep (gemm_p3.cuh 256x128) vs ping-pong (gemm_p4.cuh 256x128 three
# sets, 256x256 one set) at the ViT fc1 shape and the scoring shape, + the top-k threshold passes.  usage: bash tools/pmc_p4.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/p4_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pixelrec_amd import ops, synth
dev = "cuda"
for (m, n, k) in ((69344, 3072, 768), (69344, 768, 3072)):
    x = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev) * 0.03; b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev)
    xp, Wp = ops.split_planes(x), ops.split_planes(W)
    for th in (825612820, 425612833, 425625631):
        for _ in range(3): ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b, tile_hint=th)
    del x, W, y, xp, Wp
N, D, B = 400001, 512, 1024
table = torch.randn(N, D, device=dev) * 0.02; users = torch.randn(B, D, device=dev)
_, hu, hi, _ = synth.eval_batch(N, B, 50, np.random.default_rng(0), synth.ZipfItems(N))
ptr, items = ops.history_csr(torch.from_numpy(hu), torch.from_numpy(hi), B, dev)
tp, vmax = ops.split_planes(table), ops.row_norm_max(table)
for prod in ("6", "3", "1"):
    os.environ["PXR_TOPK_PRODUCTS"] = prod
    for _ in range(3): ops.score_topk(users, D, B, table, 10, ptr, items, table_planes=tp, table_norm_max=vmax)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT -o a -- python /tmp/p4_one.py > $OUT/stdout_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o b -- python /tmp/p4_one.py > $OUT/stdout_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT -o c -- python /tmp/p4_one.py > $OUT/stdout_c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o d -- python /tmp/p4_one.py > $OUT/stdout_d.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o e -- python /tmp/p4_one.py > $OUT/stdout_e.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o f -- python /tmp/p4_one.py > $OUT/stdout_f.log 2>&1
find $OUT -name "*.db" -delete
python $REPO/tools/pmc_summarise.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv") > $OUT/summary.log 2>&1
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in d.items():
    if not any(t in k for t in ("gemm_p3_kernel", "score_thresh")): continue
    simd_cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024          # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cyc if simd_cyc else 0
    clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / v["avg_us_under_pmc"] / 1e3 if v.get("avg_us_under_pmc") else 0
    print(f"{k[:70]:70s} us {v['avg_us_under_pmc']:8.1f}  clk {clk:4.2f} GHz  mfma_busy {busy:5.3f}  wait_any/wave_cyc {v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):5.3f}  vmem_level {v.get('SQ_INST_LEVEL_VMEM',0)/max(v.get('GRBM_GUI_ACTIVE',1)/8,1):6.1f}  hbm MB {v.get('hbm_bytes_per_launch',0)/1e6:8.1f}")
PY

#!/bin/bash
# SQ / LDS / MFMA counters of the fused tower attention kernels (forward, backward dq, backward dkv) at the PixelNet shape.
# Output: gpurun_out/pmc_tower_attn/summary.json
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_tower_attn
rm -rf "$OUT" && mkdir -p "$OUT"
cat > /tmp/ta_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
n, heads, T, d = 352, 12, 197, 64
H = heads * d
qkv = torch.randn(n * T, 3 * H, device="cuda")
dctx = torch.randn(n * T, H, device="cuda")
for _ in range(3):
    ctx, pl, lse = ops.tower_attn_fwd(qkv, n, T, heads, d, 2 * H, 0, H, d ** -0.5, ctx=True, planes=True, lse=True)
    ops.tower_attn_bwd(qkv, dctx, ctx.view(n * T, H), lse, n, T, heads, d, 2 * H, 0, H, d ** -0.5)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT" -o a -- python /tmp/ta_one.py > "$OUT/stdout_a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT" -o b -- python /tmp/ta_one.py > "$OUT/stdout_b.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d "$OUT" -o c -- python /tmp/ta_one.py > "$OUT/stdout_c.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT" -o $c -- python /tmp/ta_one.py > "$OUT/${c}_stdout.log" 2>&1
done
find "$OUT" -name "*.db" -delete
python $REPO/tools/pmc_summarise.py "$OUT/summary.json" $(find "$OUT" -name "*counter_collection.csv") > "$OUT/summary.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
tail -c 1500 "$OUT/summary.log"

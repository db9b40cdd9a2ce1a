#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/attn_pmc
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/attn_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
B, L, H, d = 64, 50, 4, 128
qkv = torch.randn(B, L, 3 * H * d, device="cuda"); km = torch.ones(B, L, dtype=torch.int64, device="cuda")
for _ in range(5):
    ctx, probs = ops.attn_fwd(qkv, km, L, B, H, L, d)
    dq = ops.attn_bwd(torch.randn_like(ctx), qkv, probs, B, H, L, d)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT -o pmc -- python /tmp/attn_one.py > $OUT/stdout.log 2>&1
rm -f $OUT/*.db
ls $OUT

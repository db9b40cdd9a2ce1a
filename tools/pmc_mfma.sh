#!/bin/bash
# MFMA utilisation (rocprofv3 PMC, own pass -- no trace domains besides --kernel-trace) of the scoring GEMM, the fused
# scoring+top-k kernel and the training-step GEMMs.  Output: gpurun_out/pmc_mfma/
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_mfma
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/mfma_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
N, D, dev = 400001, 512, "cuda"
table = torch.randn(N, D, device=dev) * 0.02
users = torch.randn(1024, D, device=dev)
scores = torch.empty(1024, N, device=dev)
for _ in range(3): ops.gemm(True, True, 1024, N, D, users, D, table, D, scores, N, ops.EPI_NONE, use_ws=False)
for _ in range(3): ops.score_topk(users, D, 1024, table, 10)
x = torch.randn(3200, 512, device=dev); W = torch.randn(1536, 512, device=dev); b = torch.randn(1536, device=dev)
y = torch.empty(3200, 1536, device=dev)
for _ in range(3): ops.gemm(True, True, 3200, 1536, 512, x, 512, W, 512, y, 1536, ops.EPI_BIAS, bias=b, use_ws=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT -o mfma -- python /tmp/mfma_one.py > $OUT/stdout.log 2>&1
rm -f $OUT/*.db
ls $OUT

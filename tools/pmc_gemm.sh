#!/bin/bash
# SQ stall / LDS counters of the training-step GEMM (64x64 tiles) -- diagnostic pass.  Output: gpurun_out/pmc_gemm/
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_gemm
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/gemm_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
dev = "cuda"
for (n, k) in ((1536, 512), (512, 512)):
    x = torch.randn(3200, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    y = torch.empty(3200, n, device=dev)
    for _ in range(3): ops.gemm(True, True, 3200, n, k, x, k, W, k, y, n, ops.EPI_BIAS, bias=b, use_ws=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT -o a -- python /tmp/gemm_one.py > $OUT/stdout_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o b -- python /tmp/gemm_one.py > $OUT/stdout_b.log 2>&1
rm -f $OUT/*.db
ls $OUT

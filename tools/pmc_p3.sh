#!/bin/bash
# SQ stall / LDS / MFMA counters of the planes GEMM.  usage: bash tools/pmc_p3.sh <outdir-name> <tile> [<tile> ...]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
shift
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/p3_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
dev = "cuda"
tiles = [int(t) for t in sys.argv[1:]]
for (m, n, k) in ((3200, 1536, 512), (3200, 512, 512), (1024, 400001, 512)):
    x = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev)
    xp, Wp = ops.split_planes(x), ops.split_planes(W)
    for th in tiles:
        for _ in range(3): ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b, tile_hint=th)
    for _ in range(3): ops.gemm(True, True, m, n, k, x, k, W, k, y, n, ops.EPI_BIAS, bias=b, use_ws=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT -o a -- python /tmp/p3_one.py "$@" > $OUT/stdout_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o b -- python /tmp/p3_one.py "$@" > $OUT/stdout_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT -o c -- python /tmp/p3_one.py "$@" > $OUT/stdout_c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o d -- python /tmp/p3_one.py "$@" > $OUT/stdout_d.log 2>&1
rm -f $OUT/*.db $OUT/*/*.db
find $OUT -name "*counter_collection.csv" | head
python $REPO/tools/pmc_summarise.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv") > $OUT/summary.log 2>&1
tail -c 3000 $OUT/summary.log

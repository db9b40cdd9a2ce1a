#!/bin/bash
# SQ / LDS / TCC counters of the grouped weight-gradient kernels (planes and in-loop split).  usage: bash tools/pmc_dw.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/dw_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
dev = "cuda"
T, D = 3200, 512
g = torch.Generator(device=dev).manual_seed(2)
probs, pl = [], []
for _ in range(2):
    for N, K in ((D, 2 * D), (2 * D, D), (D, D), (3 * D, D)):
        dy = torch.randn(T, N, device=dev, generator=g) * 0.01
        x = torch.randn(T, K, device=dev, generator=g)
        probs.append((dy, x, torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
        pl.append((ops.split_planes(dy), ops.split_planes(x), torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
for _ in range(3):
    ops.grouped_linear_bwd_weight(probs)
    for th in (812812830, 412812831, 406406431):
        ops.grouped_dw_planes(pl, tile_hint=th)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT -o a -- python /tmp/dw_one.py > $OUT/stdout_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o b -- python /tmp/dw_one.py > $OUT/stdout_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT -o c -- python /tmp/dw_one.py > $OUT/stdout_c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o d -- python /tmp/dw_one.py > $OUT/stdout_d.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o e -- python /tmp/dw_one.py > $OUT/stdout_e.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o f -- python /tmp/dw_one.py > $OUT/stdout_f.log 2>&1
rm -f $OUT/*.db $OUT/*/*.db
python $REPO/tools/pmc_summarise.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv") > $OUT/summary.log 2>&1
tail -c 300 $OUT/summary.log

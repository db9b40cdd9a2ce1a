#!/bin/bash
# HBM traffic (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE as MI355X_MICROARCH.md prescribes) of the
# HBM-bound kernels + the scoring kernels.  Output: gpurun_out/pmc_traffic/{fetch,write}_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_traffic
rm -rf $OUT && mkdir -p $OUT
cat > /tmp/traffic_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
N, D, dev = 400001, 512, "cuda"
table = torch.randn(N, D, device=dev) * 0.02
idx = torch.randint(1, N, (208896,), device=dev)
for _ in range(3): ops.embed_gather(table, idx)
tm, tv = torch.zeros_like(table), torch.zeros_like(table)
slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
for _ in range(3): ops.adamw_table(table, tm, tv, slot, None, 1e-4, 0.9, 0.999, 1e-8, 0.1, 1)
users = torch.randn(1024, D, device=dev)
scores = torch.empty(1024, N, device=dev)
for _ in range(2): ops.gemm(True, True, 1024, N, D, users, D, table, D, scores, N, ops.EPI_NONE, use_ws=False)
for _ in range(2): ops.score_topk(users, D, 1024, table, 10)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python /tmp/traffic_one.py > $OUT/fetch_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python /tmp/traffic_one.py > $OUT/write_stdout.log 2>&1
rm -f $OUT/*.db
ls $OUT

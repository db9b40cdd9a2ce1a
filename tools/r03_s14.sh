#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export PYTHONUNBUFFERED=1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --no-gemm-events --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['ms_per_step'],4))"; }
run default A=1
run dw_412812830 PXR_P3_DW_TILE=412812830
run dw_812812830 PXR_P3_DW_TILE=812812830
run dw_412806440 PXR_P3_DW_TILE=412806440
run dw_406406460 PXR_P3_DW_TILE=406406460
run dw_406406430 PXR_P3_DW_TILE=406406430
run default2 A=1

#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
SHAPES=fwd_qkv,fwd_o,fwd_f1,fwd_f2,dx_f2,dx_f1,dx_o,dx_qkv,B512_fwd_qkv,scoring,ragged,ragged_dx,o_k64 timeout 900 python tools/p3_sweep.py > $OUT/p3_sweep.log 2>&1
python - <<'PY'
import json, re
for line in open("/root/repo/gpurun_out/r03f/p3_sweep.log"):
    m = re.match(r'^(\w+) (\{.*\})$', line.strip())
    if not m:
        if "rror" in line: print(line[:300])
        continue
    name, row = m.group(1), json.loads(m.group(2))
    print(name, "b3", row["b3_default"], "best", row.get("best"))
    for k, v in row.items():
        if isinstance(v, dict) and "us" in v and k != "best": print("    ", k, v)
        elif isinstance(v, str): print("    ", k, v[:160])
PY

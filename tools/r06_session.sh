#!/bin/bash
# Round-6 measurement stages (one MI355X through gpurun).  usage: bash tools/r06_session.sh <outdir-name> [stage ...]
#   t:<pytest args>   a pytest selection (e.g. t:tests/test_gpu_bench_ranks.py)     tests   full GPU suite (-v log kept)
#   bench             the default bench line                                         prof    rocprofv3 kernel stats + timeline of the B=64 step
#   b2048             rocprofv3 kernel stats at B = 2048                             pmc64   FETCH_SIZE / WRITE_SIZE passes of the B=64 step
#   pixel / pixelprof PixelNet line / its kernel stats                               quick   bench.py --no-extras --no-cpu-baseline (headline only)
#   attn              tools/attn_bench.py                                            bq:<B>  short bench at batch B (headline fields only)
#   flake:<n>         n more full-suite runs with faulthandler, one log per run      eval    tools/eval_bench.py
#   fp64              tests/diag_fp64_trajectory.py (FP64_PERMS=<n> permuted repeats per arithmetic)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$REPO/gpurun_out/$NAME
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
cd "$REPO"
for ST in "$@"; do
case "$ST" in
t:*)
  SEL=${ST#t:}; TAG=$(echo "$SEL" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-60)
  ( time timeout 1500 python -m pytest $SEL -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -25 ) > "$OUT/pytest_$TAG.log" 2>&1
  tail -n 12 "$OUT/pytest_$TAG.log" ;;
tests)
  ( time timeout 2400 python -m pytest tests -m gpu -v 2>&1 | grep -v "amdgpu.ids\|socket.cpp" ) > "$OUT/pytest_gpu_full.log" 2>&1
  grep -v "PASSED\|^$" "$OUT/pytest_gpu_full.log" | cut -c1-400 | tail -n 60 > "$OUT/pytest_gpu.log"
  grep -c PASSED "$OUT/pytest_gpu_full.log"; tail -n 30 "$OUT/pytest_gpu.log" ;;
flake:*)
  NRUN=${ST#flake:}; mkdir -p "$OUT/flake"
  ulimit -c 0
  for i in $(seq 1 "$NRUN"); do
    ( time PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 \
        | grep -v "amdgpu.ids\|socket.cpp" | tail -n 40 ) > "$OUT/flake/run_$i.log" 2>&1
    echo "run $i: $(grep -E 'passed|failed|error|Fatal|core dumped|Aborted|Segmentation' "$OUT/flake/run_$i.log" | tail -n 2 | tr '\n' ' ')"
  done ;;
quick)
  timeout 300 python bench.py --no-extras --no-cpu-baseline > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
  python - "$OUT/bench_quick.json" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("quick:", round(d["value"]), "seq/s", round(d["ms_per_step"],4), "ms  gemm us", round(d["roofline"]["gemm_time_per_step_us"],1), "frac", round(d["roofline"]["frac"],3))
P
  ;;
bench)
  PXR_BENCH_EXTRAS="$OUT/bench_extras.json" timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  tail -n 1 "$OUT/bench_default.json" | wc -c; tail -n 1 "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err" ;;
driver)
  # exactly the driver's command (its 22.6 KB line came back unparsed)
  PXR_BENCH_EXTRAS="$OUT/bench_driver_extras.json" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"
  tail -n 1 "$OUT/bench_driver.json" | wc -c; tail -n 1 "$OUT/bench_driver.json" ;;
eval)
  timeout 400 python tools/eval_bench.py > "$OUT/eval_bench.log" 2>&1; tail -20 "$OUT/eval_bench.log" ;;
evalprof)
  P=$OUT/prof_eval; mkdir -p "$P"
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o eval -- python $REPO/tools/eval_bench.py > "$P/stdout.log" 2>&1 )
  rm -f "$P/eval_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -14 "$P/eval_kernel_stats.csv" | cut -c1-200 ;;
pixel)
  timeout 400 python bench.py --model pixelnet --no-cpu-baseline > "$OUT/bench_pixelnet_b16.json" 2> "$OUT/bench_pixelnet_b16.err"; cut -c1-600 "$OUT/bench_pixelnet_b16.json"; tail -3 "$OUT/bench_pixelnet_b16.err" ;;
prof)
  P=$OUT/prof_bench; mkdir -p "$P"
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --age-steps ${PROF_AGE:-620} --no-cpu-baseline --no-extras --no-gemm-events > "$P/bench_stdout.log" 2>&1 )
  python $REPO/tools/step_timeline.py "$P/bench_kernel_trace.csv" "$OUT/step_timeline.txt" > "$P/timeline.log" 2>&1
  rm -f "$P/bench_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -48 "$P/bench_kernel_stats.csv" | cut -c1-220 ;;
b2048)
  P=$OUT/prof_b2048; mkdir -p "$P"
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o b2048 -- python $REPO/bench.py --batch 2048 --steps 10 --warmup 3 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events > "$P/stdout.log" 2>&1 )
  rm -f "$P/b2048_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -40 "$P/b2048_kernel_stats.csv" | cut -c1-220 ;;
pixelprof)
  P=$OUT/prof_pixelnet; mkdir -p "$P"
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o pix -- python $REPO/bench.py --model pixelnet --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-gemm-events > "$P/stdout.log" 2>&1 )
  rm -f "$P/pix_kernel_trace.csv"; find "$P" -name "*.db" -delete
  head -25 "$P/pix_kernel_stats.csv" | cut -c1-220 ;;
fp64)
  timeout 600 python tests/diag_fp64_trajectory.py ${FP64_PERMS:-0} > "$OUT/fp64_trajectory.log" 2>&1; grep -v "amdgpu.ids" "$OUT/fp64_trajectory.log" | tail -24 ;;
attn)
  timeout 300 python tools/attn_bench.py > "$OUT/attn_bench.log" 2>&1; cat "$OUT/attn_bench.log" ;;
bq:*)
  BB=${ST#bq:}
  timeout 400 python bench.py --batch $BB --steps 20 --warmup 5 --age-steps 40 --no-cpu-baseline --no-extras > "$OUT/bench_b$BB.json" 2> "$OUT/bench_b$BB.err"
  python - "$OUT/bench_b$BB.json" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("B", d["config"].get("batch_per_gpu"), round(d["value"]), "seq/s", round(d["ms_per_step"],4), "ms  gemm us", round(d["roofline"]["gemm_time_per_step_us"],1))
P
  ;;
pmc64)
  Q=$OUT/pmc64; mkdir -p "$Q"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > "$Q/${c}_stdout.log" 2>&1 )
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$Q/summary.json" $(ls "$Q"/*counter_collection.csv) > /dev/null 2>&1
  python $REPO/tools/gemm_traffic.py "$Q/summary.json" "$OUT/gemm_traffic_summary.json" "round 5 code, default operands (fp16 two-plane), B=64" > "$OUT/gemm_traffic.log" 2>&1
  rm -f "$Q"/*kernel_trace.csv "$Q"/*counter_collection.csv
  cat "$OUT/gemm_traffic.log" ;;
gln)
  timeout 600 python tools/gather_ln_bench.py $GLN_NT 2>/dev/null | grep '^{' > "$OUT/gather_ln_bench.jsonl"
  python - "$OUT/gather_ln_bench.jsonl" <<'P'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("nt",d["nt"],"B",d["B"],d["ids"],d["planes"],"p",d.get("p_drop"),"xhat",d.get("xhat"),round(d["us"],1),"us",round(d["GBps"]),"GB/s",round(d["frac_of_8TBps"],3))
P
  ;;
pmcsq)
  # SQ / MFMA / LDS counters of the B = 64 step's kernels (the small h2 tiles that dominate it): three passes of 8 SQ counters
  Q=$OUT/pmcsq; mkdir -p "$Q"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$Q" -o p$i -- python $REPO/bench.py --steps 10 --warmup 2 --age-steps 20 --no-cpu-baseline --no-extras --no-gemm-events --no-graph > "$Q/p${i}_stdout.log" 2>&1 )
  done
  find "$Q" -name "*.db" -delete
  python $REPO/tools/pmc_summarise.py "$OUT/sq_counters_b64_summary.json" $(find "$Q" -name "*counter_collection.csv") > /dev/null 2>&1
  rm -rf "$Q"
  python - "$OUT/sq_counters_b64_summary.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("avg_us_under_pmc",0)*kv[1].get("launches",0))[:14]:
    cyc=v.get("GRBM_GUI_ACTIVE",0)/8; simd=cyc*1024
    print(f"{k[:72]:72s} us {v['avg_us_under_pmc']:7.1f} n {v['launches']:4d} mfma_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/simd if simd else 0:5.3f} wait_any/wave {v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):5.3f} lds_conf/lds_active {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):5.3f}")
P
  ;;
pmcgather)
  # HBM / fabric traffic of the gather as the step runs it (ln_fwd_kernel<GATHER>) and of the standalone gather, by FETCH_SIZE / WRITE_SIZE
  Q=$OUT/pmcgather; mkdir -p "$Q"
  cat > /tmp/gather_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pixelrec_amd import ops, synth
dev = "cuda"; N, D, L = 400_001, 512, 50
table = torch.randn(N, D, device=dev) * 0.02; pos = torch.randn(L, D, device=dev) * 0.02
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
zipf = synth.ZipfItems(N, seed=2020); rng = np.random.default_rng(0)
for B in (512, 2048):
    idx = torch.from_numpy(synth.train_batch(N, B, L, rng, zipf)[0][:, 0, :L].copy()).to(dev)
    for _ in range(4): ops.input_ln_fwd(table, idx, L, B, L, pos, g, b, 1e-12, 0.1, 12345, 0, save=True, planes="h2")
ids = torch.randint(1, N, (2048 * 102,), device=dev)
for _ in range(4): ops.embed_gather(table, ids)
torch.cuda.synchronize()
PY
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$Q" -o $c -- python /tmp/gather_one.py > "$Q/${c}_stdout.log" 2>&1 )
  done
  find "$Q" -name "*.db" -delete
  python - "$OUT/gather_traffic_summary.json" $(ls "$Q"/*counter_collection.csv) <<'P'
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ln_fwd_kernel" in k or "embed_gather" in k:
            acc[(k.split("(")[0].replace("void ", ""), int(r["Grid_Size"]) if "Grid_Size" in r else 0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (k, grid), cs in acc.items():
    f, w = (sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)), (sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1))
    out[f"{k} grid {grid}"] = {"fetch_kb": f, "write_kb": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                               "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads)"}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items(): print(k[:90], round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB")
P
  rm -rf "$Q" ;;
pmcp4)
  timeout 1200 bash tools/pmc_p4.sh $NAME/pmc_p4 > "$OUT/pmc_p4.log" 2>&1; tail -n 30 "$OUT/pmc_p4.log" ;;
ab:*)
  # ab:<label>:<ENV=VAL,ENV=VAL|->:<bench args with + for spaces>   short headline-only run under the given knobs
  IFS=: read -r _ LAB ENVS ARGS <<< "$ST"
  ARGS=$(echo "$ARGS" | tr '+' ' ')
  ( [ "$ENVS" != "-" ] && export $(echo "$ENVS" | tr ',' ' ')
    PXR_BENCH_EXTRAS="$OUT/ab_${LAB}_extras.json" timeout 400 python bench.py --no-cpu-baseline --no-extras $ARGS > "$OUT/ab_$LAB.json" 2> "$OUT/ab_$LAB.err" )
  python - "$OUT/ab_$LAB.json" "$LAB" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ab", sys.argv[2], "B", d["config"].get("batch_per_gpu"), round(d["value"]), "seq/s", round(d["ms_per_step"],4), "ms  host", round(d.get("host_enqueue_ms_per_step",0),4), "gemm us", round(d["roofline"].get("gemm_time_per_step_us",0),1), "graph", d["config"].get("hip_graph"))
except Exception as e:
    print("ab", sys.argv[2], "FAILED", e)
P
  tail -n 2 "$OUT/ab_$LAB.err" | cut -c1-300 ;;
*) echo "unknown stage $ST" ;;
esac
done

#!/bin/bash
# Step timelines (tools/step_timeline.py) of the default bench: default schedule and PXR_SORT_OVERLAP=0.  GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/timeline_ab
mkdir -p "$O"
for v in new serial; do
  if [ $v = serial ]; then export PXR_SORT_OVERLAP=0; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O" -o tr_$v -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-gemm-events > "$O/stdout_$v.log" 2>&1
  python $R/tools/step_timeline.py "$O/tr_${v}_kernel_trace.csv" "$O/step_timeline_$v.txt" > "$O/tl_$v.log" 2>&1; ls -la "$O" >> "$O/tl_$v.log"
  rm -f "$O/tr_${v}_kernel_trace.csv"
done
find "$O" -name "*.db" -delete

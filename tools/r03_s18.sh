#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03q
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pixelnet -- python $REPO/bench.py --model pixelnet --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pixelnet_stdout.log 2>&1
rm -f $OUT/*.db $OUT/*kernel_trace.csv $OUT/*agent_info.csv
ls $OUT

#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02g
mkdir -p $OUT
cd $REPO
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-extras"
timeout 300 python bench.py $B --force-collectives > $OUT/bench_fc.json 2> $OUT/bench_fc.err
timeout 300 python bench.py $B --force-collectives --full-exchange > $OUT/bench_fc_full.json 2> $OUT/bench_fc_full.err
timeout 300 python bench.py --model pixelnet --steps 10 --warmup 3 > $OUT/bench_pixelnet_b16.json 2> $OUT/bench_pixelnet_b16.err
timeout 300 python bench.py --model pixelnet --encoder clip-vit-base-patch32 --steps 10 --warmup 3 > $OUT/bench_pixelnet_b32.json 2> $OUT/bench_pixelnet_b32.err
tail -n 25 $OUT/pytest_gpu.log
for f in bench_fc bench_fc_full bench_pixelnet_b16 bench_pixelnet_b32; do echo == $f; tail -c 1500 $OUT/$f.json; tail -n 3 $OUT/$f.err; done

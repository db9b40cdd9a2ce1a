#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/run_gpu_tests.sh [pytest args]
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q "$@" 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40

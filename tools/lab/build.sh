#!/bin/bash
# builds the lab executables next to their sources (git-ignored; they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")"
for f in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -fno-gpu-rdc \
        -Rpass-analysis=kernel-resource-usage "$f.hip" -o "$f.bin" 2> "$f.build.log" || { tail -30 "$f.build.log"; exit 1; }
  grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" "$f.build.log" | paste - - - - - - - - | sed 's/remark: [^ ]*: //g' | cut -c1-400 > "$f.resources.txt" || true
done

#!/bin/bash
set -e
mkdir -p gpurun_out/r02
: > gpurun_out/r02/prepare_tile_us.json
for rows in 2 3 4 6; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Igcn10_amd/csrc -DGCN10_EXPAND_ROWS=$rows -shared -o gcn10_amd/libgcn10_gpu.so gcn10_amd/csrc/gcn10_deflate.hip gcn10_amd/csrc/gcn10_deflate_fused.hip gcn10_amd/csrc/gcn10_gpu.hip gcn10_amd/csrc/gcn10_inflate.hip 2>/dev/null
  echo "rows per thread $rows" >> gpurun_out/r02/prepare_tile_us.json
  python3 tools/time_prepare_tile.py 36000 >> gpurun_out/r02/prepare_tile_us.json 2>&1
done
cat gpurun_out/r02/prepare_tile_us.json

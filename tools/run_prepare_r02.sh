#!/bin/bash
set -e
mkdir -p gpurun_out/r02
python3 tools/time_prepare_tile.py 36000 > gpurun_out/r02/prepare_tile_us.json 2>&1
cat gpurun_out/r02/prepare_tile_us.json
GCN10_BENCH_AB=1 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > gpurun_out/r02/bench_ab.json 2> gpurun_out/r02/bench_ab.err

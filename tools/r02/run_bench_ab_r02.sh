#!/bin/bash
# bench with the in-process A/B diagnostic (per-step kernel times of the timed region; the same launch with and
# without gcn10_gpu_prepare_tile in front, afterwards)
set -e
mkdir -p gpurun_out/r02
GCN10_BENCH_AB=1 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > gpurun_out/r02/bench_ab.json 2> gpurun_out/r02/bench_ab.err
GCN10_BENCH_AB=1 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --pre-warm-ms 0 > gpurun_out/r02/bench_ab_nowarm.json 2> gpurun_out/r02/bench_ab_nowarm.err
echo done

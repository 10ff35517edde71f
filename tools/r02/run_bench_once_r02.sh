#!/bin/bash
mkdir -p gpurun_out/r02
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_once.json 2> gpurun_out/r02/bench_once.err
tail -c 600 gpurun_out/r02/bench_once.err
echo done

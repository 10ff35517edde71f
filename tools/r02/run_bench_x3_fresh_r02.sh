#!/bin/bash
# the driver's bench command three times, each a fresh process; the first is the first GPU process of the box
mkdir -p gpurun_out/r02
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_fresh_$i.json 2> gpurun_out/r02/bench_fresh_$i.err
done
echo done

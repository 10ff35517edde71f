#!/bin/bash
# two ranks sharing the one GPU of the box (rehearsal of the N > 1 path: launcher, file rendezvous, per-rank records)
set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python3 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 5 > gpurun_out/r02/bench_n2_oversubscribed.json 2> gpurun_out/r02/bench_n2_oversubscribed.err
tail -c 1500 gpurun_out/r02/bench_n2_oversubscribed.json
timeout -k 10 600 python3 bench.py --gpus 2 --oversubscribe --scaling strong --steps 20 --warmup 5 > gpurun_out/r02/bench_n2_strong_oversubscribed.json 2> gpurun_out/r02/bench_n2_strong_oversubscribed.err
echo done

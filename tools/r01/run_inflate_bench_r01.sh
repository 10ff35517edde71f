#!/bin/bash
set -e
for p in patches natural iid; do
  timeout -k 10 280 python tools/bench_inflate.py --pattern $p > gpurun_out/inflate_$p.json
  cat gpurun_out/inflate_$p.json
done

#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --keep --esa-compression 8 --strip-rows 1024 > gpurun_out/sr_1024.json
for sr in 0 1536 1792 2048; do
  timeout -k 10 300 python tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --reuse --keep --esa-compression 8 --strip-rows $sr > gpurun_out/sr_$sr.json
done
for sr in 1024 0 1536 1792 2048; do python3 -c "
import json; d=json.load(open('gpurun_out/sr_$sr.json')); m=d['modes']['files']; print('strip_rows $sr:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'])"; done
timeout -k 10 600 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -2

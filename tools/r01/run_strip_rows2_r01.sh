#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern natural --blocks 12 --modes files --keep --esa-compression 8 --strip-rows 1024 > gpurun_out/srn_a.json
for rep in 1 2; do for sr in 1024 1792 768 1280; do
  timeout -k 10 300 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes files --reuse --keep --esa-compression 8 --strip-rows $sr > gpurun_out/srn_$sr.json
  python3 -c "
import json; d=json.load(open('gpurun_out/srn_$sr.json')); m=d['modes']['files']; print('natural strip_rows $sr rep $rep:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'])"
done; done

#!/bin/bash
set -e
for pat in patches natural; do for f in -1 0.1; do
  timeout -k 10 600 python tools/bench_pipeline.py --pattern $pat --blocks 12 --modes files --esa-compression 8 --dual-soil-fraction $f > gpurun_out/alias_${pat}_$f.json
  python3 -c "
import json; d=json.load(open('gpurun_out/alias_${pat}_$f.json')); m=d['modes']['files']; print('$pat dual fraction $f:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], m['output_bytes'])"
done; done

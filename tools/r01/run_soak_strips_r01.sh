#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern patches --blocks 4 --repeat 20 --modes files --keep --esa-compression 8 > gpurun_out/ss_base.json
for sr in 1024 1792 1024 1792 2560; do
  python tools/bench_pipeline.py --pattern patches --blocks 4 --repeat 20 --modes files --reuse --keep --esa-compression 8 --strip-rows $sr > gpurun_out/ss_$sr.json
  python3 -c "
import json; d=json.load(open('gpurun_out/ss_$sr.json')); m=d['modes']['files']; print('patches strip_rows $sr: after start-up', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'])"
done

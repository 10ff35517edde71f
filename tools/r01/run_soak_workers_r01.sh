#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern patches --blocks 4 --repeat 20 --modes files --keep --esa-compression 8 > gpurun_out/sw_base.json
for wk in 1 2 3 4; do
  python tools/bench_pipeline.py --pattern patches --blocks 4 --repeat 20 --modes files --reuse --keep --esa-compression 8 --workers-per-gpu $wk > gpurun_out/sw_$wk.json
  python3 -c "
import json; d=json.load(open('gpurun_out/sw_$wk.json')); m=d['modes']['files']; print('workers $wk: after start-up', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'])"
done

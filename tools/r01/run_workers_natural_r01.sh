#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern natural --blocks 12 --modes files --keep --esa-compression 8 > gpurun_out/wn_base.json
for wk in 2 3 4; do
  timeout -k 10 300 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes files --reuse --keep --esa-compression 8 --workers-per-gpu $wk > gpurun_out/wn_$wk.json
  python3 -c "
import json; d=json.load(open('gpurun_out/wn_$wk.json')); m=d['modes']['files']; print('natural workers $wk:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], m['worker_seconds'])"
done
timeout -k 10 300 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes null --reuse --esa-compression 8 > gpurun_out/wn_null.json
python3 -c "
import json; d=json.load(open('gpurun_out/wn_null.json')); m=d['modes']['null']; print('natural null sink:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], m['worker_seconds'])"

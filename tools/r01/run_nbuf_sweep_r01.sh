#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern patches --blocks 24 --modes files --keep --esa-compression 8 > gpurun_out/sweep_build.json
for nb in 2 3 4; do for wk in 2 3; do
  GCN10_STRIP_BUFFERS=$nb timeout -k 10 300 python tools/bench_pipeline.py --pattern patches --blocks 24 --modes files --reuse --keep --esa-compression 8 --workers-per-gpu $wk > gpurun_out/sweep_${nb}_${wk}.json
  python3 -c "
import json; d=json.load(open('gpurun_out/sweep_${nb}_${wk}.json')); m=d['modes']['files']; print('nbuf $nb workers $wk:', m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], '|', m['worker_seconds'][:95])"
done; done

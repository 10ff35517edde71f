#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern patches --blocks 32 --modes null,files,null --esa-compression 8 > gpurun_out/null32.json
python3 -c "
import json; d=json.load(open('gpurun_out/null32.json'))
for k,m in d['modes'].items(): print(k, m['seconds'], m['seconds_per_block'], '| after start-up', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'], '|', m['worker_seconds'][:110])"
df -h /tmp | tail -1; mount | grep " /tmp " | head -2

#!/bin/bash
# long runs on a small world: 4 block geometries listed R times (first-use allocations amortised)
set -e
for R in 10 40; do
python tools/bench_pipeline.py --pattern patches --blocks 4 --repeat $R --modes files --esa-compression 8 > gpurun_out/soak_$R.json
python3 -c "
import json; d=json.load(open('gpurun_out/soak_$R.json'))
for k,m in d['modes'].items(): print('repeat $R', k, 'blocks', m['blocks_done'], 'wall', m['seconds'], 'after start-up', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'], 'rc', m['rc'])"
done

#!/bin/bash
set -e
python tools/bench_pipeline.py --pattern natural --blocks 4 --repeat 10 --modes null --esa-compression 8 > gpurun_out/soak_natural_null.json
python tools/bench_pipeline.py --pattern natural --blocks 4 --repeat 5 --modes files --esa-compression 8 > gpurun_out/soak_natural_files.json
for f in soak_natural_null soak_natural_files; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json'))
for k,m in d['modes'].items(): print('$f', k, 'blocks', m['blocks_done'], 'wall', m['seconds'], 'after start-up', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'], 'rc', m['rc'])"; done

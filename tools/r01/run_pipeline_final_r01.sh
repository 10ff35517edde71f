#!/bin/bash
# the pipeline numbers quoted in DESIGN.md section 6, one call
set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 > gpurun_out/final_patches_deflate32.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 16 --modes files --esa-compression 8 > gpurun_out/final_natural_deflate16.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 16 --modes files,null --esa-compression 1 > gpurun_out/final_patches_raw16.json
for f in final_patches_deflate32 final_natural_deflate16 final_patches_raw16; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json'))
for k,m in d['modes'].items(): print('$f', k, m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], '| after start-up:', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'])"; done

#!/bin/bash
# the pipeline numbers of round 2 (same commands as tools/r01/run_pipeline_final_r01.sh), one call
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 > gpurun_out/r02_final_patches_deflate32.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 16 --modes files,null --esa-compression 8 > gpurun_out/r02_final_natural_deflate16.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 8 --modes null --esa-compression 8 --real-vrt-pixel > gpurun_out/r02_final_natural_36001px.json
for f in r02_final_patches_deflate32 r02_final_natural_deflate16 r02_final_natural_36001px; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json'))
for k,m in d['modes'].items(): print('$f', k, m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], '| after start-up:', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'])"; done

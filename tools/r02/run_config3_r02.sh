#!/bin/bash
# BASELINE config 3 at full block size: a queue of 36000^2 blocks, single lookup (g_ii, drained),
# DEFLATE landcover in, one DEFLATE GeoTIFF per block out; and the same queue with all 18 rasters
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for pat in patches natural; do
timeout -k 10 900 python tools/bench_pipeline.py --pattern $pat --blocks 24 --modes files --esa-compression 8 --lookups g_ii --conditions drained --keep --workdir /tmp/gcn10_c3 > gpurun_out/r02_config3_single_lookup_$pat.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern $pat --blocks 24 --modes files --esa-compression 8 --reuse --workdir /tmp/gcn10_c3 > gpurun_out/r02_config3_all18_$pat.json
for f in r02_config3_single_lookup_$pat r02_config3_all18_$pat; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json'))
for k,m in d['modes'].items(): print('$f', d['rasters_per_block'], k, m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s'], '| after start-up:', m['seconds_after_startup'], m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'], m['output_bytes'])"; done
done

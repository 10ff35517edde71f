#!/bin/bash
# block rate against workers per GPU (default 2): noisy landcover with and without the file sink, patchy with files
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02
for w in 2 3 4; do
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern natural --blocks 16 --modes null,files --esa-compression 8 --workers-per-gpu $w > gpurun_out/r02/pipeline_workers_${w}_natural.json
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 --workers-per-gpu $w > gpurun_out/r02/pipeline_workers_${w}_patches.json
  for f in natural patches; do python3 -c "
import json; d=json.load(open('gpurun_out/r02/pipeline_workers_${w}_$f.json'))
for k,m in d['modes'].items(): print('workers $w', '$f', k, 's/block', m['seconds_per_block'], '| after start-up:', m['steady_seconds_per_block'], m['steady_cn_gpx_per_s'])"; done
done

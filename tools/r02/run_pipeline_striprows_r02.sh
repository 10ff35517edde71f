#!/bin/bash
# block rate against rows per strip (config key strip_rows, default 768) with four strip buffer sets
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02
: > gpurun_out/r02/pipeline_strip_rows.txt
for sr in 768 512 1024 1536 768 1024; do
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 --strip-rows $sr > gpurun_out/r02/sr_p.json
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern natural --blocks 16 --modes null,files --esa-compression 8 --strip-rows $sr > gpurun_out/r02/sr_n.json
  python3 -c "
import json
for f in ('sr_p','sr_n'):
    d=json.load(open('gpurun_out/r02/%s.json'%f))
    for k,m in d['modes'].items(): print('strip_rows $sr', d['pattern'], k, 'after start-up s/block', m['steady_seconds_per_block'])" | tee -a gpurun_out/r02/pipeline_strip_rows.txt
done

#!/bin/bash
# block rate against the number of strip buffer sets per worker (GCN10_STRIP_BUFFERS, default 3)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02
echo "second call: 4 6 8 3 5 6" >> gpurun_out/r02/pipeline_strip_buffers.txt
for nb in 4 6 8 3 5 6; do
  GCN10_STRIP_BUFFERS=$nb timeout -k 10 900 python3 tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 > gpurun_out/r02/nbuf_p.json
  GCN10_STRIP_BUFFERS=$nb timeout -k 10 900 python3 tools/bench_pipeline.py --pattern natural --blocks 16 --modes null,files --esa-compression 8 > gpurun_out/r02/nbuf_n.json
  python3 -c "
import json
for f in ('nbuf_p','nbuf_n'):
    d=json.load(open('gpurun_out/r02/%s.json'%f))
    for k,m in d['modes'].items(): print('strip buffer sets $nb', d['pattern'], k, 'after start-up s/block', m['steady_seconds_per_block'])" | tee -a gpurun_out/r02/pipeline_strip_buffers.txt
done

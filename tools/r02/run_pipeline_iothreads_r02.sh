#!/bin/bash
# block rate against the size of the I/O pool (config key io_threads; the program's choice is min(64, cpus - 1))
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02
: > gpurun_out/r02/pipeline_io_threads.txt
for t in 0 8 12 16 24 32 0 16; do
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 --io-threads $t > gpurun_out/r02/iot_p.json
  timeout -k 10 900 python3 tools/bench_pipeline.py --pattern natural --blocks 16 --modes null,files --esa-compression 8 --io-threads $t > gpurun_out/r02/iot_n.json
  python3 -c "
import json
for f in ('iot_p','iot_n'):
    d=json.load(open('gpurun_out/r02/%s.json'%f))
    for k,m in d['modes'].items(): print('io_threads $t', d['pattern'], k, 'after start-up s/block', m['steady_seconds_per_block'])" | tee -a gpurun_out/r02/pipeline_io_threads.txt
done

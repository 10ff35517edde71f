#!/bin/bash
set -e
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 24 --modes files --esa-compression 8 --gpu-inflate 1 > gpurun_out/pipeline_timers.json
python3 -c "
import json; d=json.load(open('gpurun_out/pipeline_timers.json')); m=d['modes']['files']; print(m['seconds'], m['seconds_per_block'], m['cn_gpx_per_s']); print(m['worker_seconds'])"

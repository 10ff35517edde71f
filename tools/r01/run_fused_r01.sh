#!/bin/bash
# fused encoder in the pipeline: 16 and 32 full-size blocks (mode 2), 16 blocks mode 1
set -e
mkdir -p gpurun_out
timeout -k 10 500 python tools/bench_pipeline.py --blocks 16 --modes files --keep --gpu-deflate 2 > gpurun_out/pipeline_fused.json
cut -c150-520 gpurun_out/pipeline_fused.json
timeout -k 10 300 python tools/bench_pipeline.py --blocks 16 --modes null --reuse --keep --gpu-deflate 2 > gpurun_out/pipeline_fused_null.json
cut -c150-520 gpurun_out/pipeline_fused_null.json
timeout -k 10 300 python tools/bench_pipeline.py --blocks 16 --modes files --reuse --keep --gpu-deflate 2 --workers-per-gpu 3 > gpurun_out/pipeline_fused_w3.json
cut -c150-520 gpurun_out/pipeline_fused_w3.json
timeout -k 10 300 python tools/bench_pipeline.py --blocks 16 --modes files --reuse --gpu-deflate 1 > gpurun_out/pipeline_unfused.json
cut -c150-520 gpurun_out/pipeline_unfused.json

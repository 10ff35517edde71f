#!/bin/bash
# DEFLATE landcover (1024x1024 tiles, as the ESA files): GPU inflate vs host inflate, end to end
set -e
mkdir -p gpurun_out
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 16 --modes files --keep --esa-compression 8 --gpu-inflate 1 > gpurun_out/pipeline_inflate_gpu_natural.json
cut -c100-560 gpurun_out/pipeline_inflate_gpu_natural.json
timeout -k 10 300 python tools/bench_pipeline.py --pattern natural --blocks 16 --modes files --reuse --esa-compression 8 --gpu-inflate 0 > gpurun_out/pipeline_inflate_host_natural.json
cut -c100-560 gpurun_out/pipeline_inflate_host_natural.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 32 --modes files --esa-compression 8 --gpu-inflate 1 > gpurun_out/pipeline_inflate_gpu_patches32.json
cut -c100-560 gpurun_out/pipeline_inflate_gpu_patches32.json

#!/bin/bash
# fused encoder in the pipeline: CLI tests, then 16 full-size blocks in modes 2 and 1
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_cli.py tests/test_gpu_deflate.py -m gpu -x -q > gpurun_out/fused_tests.log 2>&1 || { tail -30 gpurun_out/fused_tests.log; exit 1; }
tail -2 gpurun_out/fused_tests.log
timeout -k 10 500 python tools/bench_pipeline.py --blocks 16 --modes files --keep --gpu-deflate 2 > gpurun_out/pipeline_fused.json
cat gpurun_out/pipeline_fused.json
timeout -k 10 300 python tools/bench_pipeline.py --blocks 16 --modes files --reuse --gpu-deflate 1 > gpurun_out/pipeline_unfused.json
cat gpurun_out/pipeline_unfused.json

#!/bin/bash
set -e
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 3 --modes files --esa-compression 8 --real-vrt-pixel --keep --workdir /tmp/gcn10_36001 > gpurun_out/pipeline_36001.json
cut -c1-700 gpurun_out/pipeline_36001.json
# (the check of its rasters against the oracle is tests/test_cli.py::test_full_size_block_of_the_real_vrt_shape)

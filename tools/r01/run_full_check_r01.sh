#!/bin/bash
# everything the driver runs at round end, in one call
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_now.json
cat gpurun_out/bench_now.json

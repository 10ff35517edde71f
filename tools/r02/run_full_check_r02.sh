#!/bin/bash
# everything the driver runs at round end, in one call; first the state of the box, then prepare_tile's time
set -e
mkdir -p gpurun_out/r02
PROBE_COMPACT=1 PROBE_LABEL=before python3 tools/box_state_probe.py > gpurun_out/r02/box_state_now.jsonl 2>&1 || true
python3 tools/time_prepare_tile.py 36000 > gpurun_out/r02/prepare_tile_us.json 2>&1 || true
cat gpurun_out/r02/prepare_tile_us.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_now.json
cat gpurun_out/bench_now.json
PROBE_COMPACT=1 PROBE_LABEL=after python3 tools/box_state_probe.py >> gpurun_out/r02/box_state_now.jsonl 2>&1 || true

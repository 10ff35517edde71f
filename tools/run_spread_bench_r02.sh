#!/bin/bash
# bench with the spread candidates in the calibration as the FIRST GPU process of the box, the parity test of the
# placed allocations, then the bench twice more
set -e
mkdir -p gpurun_out/r02
PROBE_COMPACT=1 PROBE_LABEL=first python3 tools/box_state_probe.py > gpurun_out/r02/box_state_now.jsonl 2>&1 || true
python3 bench.py --no-cpu-baseline --no-also > gpurun_out/r02/bench_spread_first.json 2> gpurun_out/r02/bench_spread_first.err
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placed or stream_copy or tune" 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-also > gpurun_out/r02/bench_spread_second.json 2> gpurun_out/r02/bench_spread_second.err
python3 bench.py --no-cpu-baseline --no-also > gpurun_out/r02/bench_spread_third.json 2> gpurun_out/r02/bench_spread_third.err
echo done

#!/bin/bash
# bench with the spread candidates in the calibration (as the FIRST GPU process of the box), then the new parity
# test, then the bench again (a second process on the same box) and once with plain allocations only
set -e
mkdir -p gpurun_out/r02
python3 bench.py --no-cpu-baseline > gpurun_out/r02/bench_spread_first.json 2> gpurun_out/r02/bench_spread_first.err
tail -c 3000 gpurun_out/r02/bench_spread_first.json
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placed or stream_copy or tune" 2>&1 | tail -3
python3 bench.py --no-cpu-baseline > gpurun_out/r02/bench_spread_second.json 2> gpurun_out/r02/bench_spread_second.err
python3 bench.py --no-cpu-baseline --no-spread > gpurun_out/r02/bench_spread_off.json 2> gpurun_out/r02/bench_spread_off.err
echo done

#!/bin/bash
set -e
mkdir -p gpurun_out/r02
python3 tools/time_series.py > gpurun_out/r02/time_series.jsonl 2> gpurun_out/r02/time_series.err
echo done

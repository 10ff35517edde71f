#!/bin/bash
set -e
mkdir -p gpurun_out/r02
python3 tools/time_prepare_effect.py > gpurun_out/r02/time_prepare_effect.jsonl 2> gpurun_out/r02/time_prepare_effect.err
echo done

#!/bin/bash
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/spread_lab.txt
timeout -k 10 400 tools/spread_lab 2 32 > $O 2>&1
timeout -k 10 300 tools/spread_lab 32 40 >> $O 2>&1
cat $O

#!/bin/bash
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/region_lab.txt
timeout -k 10 300 tools/region_lab 40 1 > $O 2>&1
timeout -k 10 300 tools/region_lab 40 0 >> $O 2>&1
cat $O

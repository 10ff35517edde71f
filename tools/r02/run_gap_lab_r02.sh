#!/bin/bash
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/gap_lab.txt
timeout -k 10 400 tools/gap_lab > $O 2>&1
cat $O

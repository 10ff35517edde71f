#!/bin/bash
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/slab_lab.txt
timeout -k 10 300 tools/slab_lab 1 > $O 2>&1
timeout -k 10 300 tools/slab_lab 0 >> $O 2>&1
cat $O

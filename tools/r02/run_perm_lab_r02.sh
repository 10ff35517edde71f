#!/bin/bash
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/perm_lab.txt
timeout -k 10 300 tools/perm_lab 1 > $O 2>&1
timeout -k 10 300 tools/perm_lab 0 >> $O 2>&1
cat $O

#!/bin/bash
# contig_lab as the first GPU process of the box, then with other source recipes
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/contig_lab.txt
timeout -k 10 300 tools/contig_lab 8 0 > $O 2>&1
timeout -k 10 300 tools/contig_lab 8 2 >> $O 2>&1
timeout -k 10 300 tools/contig_lab 8 1 >> $O 2>&1
timeout -k 10 300 tools/contig_lab 8 0 >> $O 2>&1
cat $O

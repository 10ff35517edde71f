#!/bin/bash
set -e
mkdir -p gpurun_out/r02
LAB_SPREAD=1 timeout -k 10 500 tools/strip_lab 36000 36000 5 5 gap > gpurun_out/r02/strip_lab_gap_spread.jsonl 2> gpurun_out/r02/strip_lab_gap_spread.err
LAB_SPREAD=0 timeout -k 10 500 tools/strip_lab 36000 36000 5 5 gap > gpurun_out/r02/strip_lab_gap_plain.jsonl 2> gpurun_out/r02/strip_lab_gap_plain.err
echo done

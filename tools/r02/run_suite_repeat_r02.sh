#!/bin/bash
# the whole GPU suite N times in a row (one process each), then smoke and the bench line
mkdir -p gpurun_out/r02
: > gpurun_out/r02/suite_repeats.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/suite_run_$i.txt 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/r02/suite_run_$i.txt)" | tee -a gpurun_out/r02/suite_repeats.txt
  grep -n "MISMATCH" gpurun_out/r02/suite_run_$i.txt | head -4 | cut -c1-200 | tee -a gpurun_out/r02/suite_repeats.txt
done
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_now.json
cat gpurun_out/bench_now.json | cut -c1-1500

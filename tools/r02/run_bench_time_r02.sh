#!/bin/bash
# how long the default bench command takes on the box (CPU baseline + calibration + measurement)
mkdir -p gpurun_out/r02
S=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_timed_whole_command.json 2> gpurun_out/r02/bench_timed_whole_command.err
E=$(date +%s.%N)
python3 -c "print(\"whole command: %.1f s\" % ($E - $S))" | tee gpurun_out/r02/bench_whole_command_seconds.txt
bash tools/r02/run_rehearse_n2_r02.sh > /dev/null 2>&1
cp gpurun_out/r02/bench_n2_oversubscribed.json gpurun_out/r02/bench_2ranks_weak_one_gpu_rehearsal.json
cp gpurun_out/r02/bench_n2_strong_oversubscribed.json gpurun_out/r02/bench_2ranks_strong_one_gpu_rehearsal.json
echo done

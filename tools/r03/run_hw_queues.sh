#!/bin/bash
# hardware queues per process (GPU_MAX_HW_QUEUES, ROCm default 4): with two workers the program has six streams
# (kernel, copy-back and input per worker); streams that share a hardware queue run in order
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_q_$pat > /dev/null 2>&1
  for rep in 1 2; do for q in 4 8 16; do
    modes=null; [ $pat = patches ] && modes=null,files
    echo -n "$pat GPU_MAX_HW_QUEUES=$q rep $rep: "
    GPU_MAX_HW_QUEUES=$q python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes $modes --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_q_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']
print(' '.join('%s %s (cpu %s)' % (m, v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block']) for m, v in d.items()))"
  done; done
  rm -rf /tmp/gcn10_q_$pat
done 2>&1 | tee $O/hw_queues_72_blocks.txt

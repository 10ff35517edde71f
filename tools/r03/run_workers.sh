#!/bin/bash
# workers per GPU with the steady-state metric (every worker's blocks after its first), 72 blocks, null sink and files (patchy)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_wk_$pat > /dev/null 2>&1
  for rep in 1 2; do for wk in 1 2 3 4; do
    modes=null; [ $pat = patches ] && modes=null,files
    echo -n "$pat workers $wk rep $rep: "
    python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes $modes --keep --reuse --workers-per-gpu $wk --esa-compression 8 --workdir /tmp/gcn10_wk_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']
print(' '.join('%s %s (cpu %s)' % (m, v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block']) for m, v in d.items()))"
  done; done
  rm -rf /tmp/gcn10_wk_$pat
done 2>&1 | tee $O/workers_per_gpu_72_blocks.txt

#!/bin/bash
# waits for GPU events: query + sleep (GCN10_EVENT_SLEEP_US, default 50) against the runtime's spinning wait (0);
# 72 blocks, steady state, alternating
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_es_$pat > /dev/null 2>&1
  for rep in 1 2; do for us in 0 50 200; do
    modes=null; [ $pat = patches ] && modes=null,files
    echo -n "$pat sleep_us=$us rep $rep: "
    GCN10_EVENT_SLEEP_US=$us python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes $modes --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_es_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']
print(' '.join('%s %s (cpu %s, user/sys %s)' % (m, v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block'], v['host_cpu_user_system']) for m, v in d.items()))"
  done; done
  rm -rf /tmp/gcn10_es_$pat
done 2>&1 | tee $O/event_sleep_72_blocks.txt

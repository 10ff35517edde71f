#!/bin/bash
# rows per strip x strip buffer sets on noisy blocks (null sink), end-of-round kernels, 72 blocks, steady state
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
pat=natural
python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_sn > /dev/null 2>&1
for rep in 1 2; do for cfg in "768 3" "768 4" "1536 4" "2304 4" "3072 4" "4096 4"; do
  set -- $cfg
  echo -n "natural strip_rows $1 buffers $2 rep $rep: "
  GCN10_STRIP_BUFFERS=$2 python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --strip-rows $1 --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_sn | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']
print(' '.join('%s %s (cpu %s, pinned %s MB, rss %s MB)' % (m, v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block'], v['pinned_MB'], v['peak_rss_MB']) for m, v in d.items()))"
done; done 2>&1 | tee $O/strip_rows3_natural_72_blocks.txt
rm -rf /tmp/gcn10_sn

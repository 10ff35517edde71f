#!/bin/bash
# blocks of input staged ahead of the encoder: one (N_IN = 2 slots, the product) against two (variants/nin3: the
# program built with N_IN = 3), 72 blocks, steady state, null sink
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_id_$pat > /dev/null 2>&1
  for rep in 1 2; do for v in one two; do
    bin=$R/bin/gcn10; [ $v = two ] && bin=$R/variants/nin3/bin/gcn10
    echo -n "$pat blocks ahead: $v rep $rep: "
    GCN10_BIN=$bin python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_id_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']['null']
print(d['after_first_block_seconds_per_block'], 'cpu', d['host_cpu_seconds_per_block'], 'pinned', d['pinned_MB'])"
  done; done
  rm -rf /tmp/gcn10_id_$pat
done 2>&1 | tee $O/input_depth_72_blocks.txt

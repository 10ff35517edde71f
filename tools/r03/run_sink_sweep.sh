#!/bin/bash
# file sink on patchy blocks with the end-of-round kernels (a strip is ~0.11 ms of GPU work now): strip buffer
# sets x drain lag x I/O threads, 72 blocks, steady state
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
pat=patches
python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_sk > /dev/null 2>&1
for rep in 1 2; do for cfg in "2 3 0" "2 4 0" "3 6 0" "5 8 0" "7 12 0" "3 6 24" "5 8 32"; do
  set -- $cfg
  echo -n "drain_lag $1 buffers $2 io_threads $3 rep $rep: "
  GCN10_DRAIN_LAG=$1 GCN10_STRIP_BUFFERS=$2 python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes files --io-threads $3 --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_sk | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']['files']
print(d['after_first_block_seconds_per_block'], 'cpu', d['host_cpu_seconds_per_block'], d['worker_seconds'][:140])"
done; done 2>&1 | tee $O/sink_sweep_patches_72_blocks.txt
rm -rf /tmp/gcn10_sk

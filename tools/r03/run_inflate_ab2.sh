#!/bin/bash
# decoder with an 8 KiB LDS ring (variants/libgcn10_gpu_w8k.so: ten streams per CU, an F-A workgroup fits beside five)
# against the default 16 KiB, in the PIPELINE, steady state after each worker's first block, 72 blocks, null sink
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_w_$pat > /dev/null 2>&1
  for rep in 1 2 3; do for v in w16k w8k; do for wk in 1 2; do
    lib=$R/gcn10_amd/libgcn10_gpu.so; [ $v = w8k ] && lib=$R/variants/libgcn10_gpu_w8k.so
    echo -n "$pat $v workers $wk rep $rep: "
    GCN10_GPU_LIB=$lib python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --reuse --workers-per-gpu $wk --esa-compression 8 --workdir /tmp/gcn10_w_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']['null']
print(d['after_first_block_seconds_per_block'])"
  done; done; done
  rm -rf /tmp/gcn10_w_$pat
done 2>&1 | tee $O/inflate_window_pipeline_72_blocks.txt

#!/bin/bash
# the program (bin/gcn10, this tree) with the GPU library of this tree ("head") and of variants/<name>/ side by side:
# steady state after each worker's first block, 72 blocks of DEFLATE landcover, alternating runs on one box
# usage: VARIANTS="bprime" MODES="null files" REPS=2 run_pipeline_variants.sh <out name>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_v_$pat > /dev/null 2>&1
  for rep in $(seq 1 ${REPS:-2}); do for v in $VARIANTS head; do for mode in ${MODES:-null}; do
    [ $mode = files ] && [ $pat = natural ] && continue      # (2.1 GB of files per noisy block: PCIe-bound, not this comparison)
    lib=$R/gcn10_amd/libgcn10_gpu.so; [ $v != head ] && lib=$R/variants/$v/libgcn10_gpu.so
    echo -n "$pat $v $mode rep $rep: "
    GCN10_GPU_LIB=$lib python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes $mode --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_v_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']['$mode']
print(d['after_first_block_seconds_per_block'], 'cpu', d['host_cpu_seconds_per_block'])"
  done; done; done
  rm -rf /tmp/gcn10_v_$pat
done 2>&1 | tee $O/${1:-variants}.txt

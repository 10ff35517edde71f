#!/bin/bash
# same-box A/B of builds of libgcn10_gpu.so (variants/libgcn10_gpu_<name>.so) on the fused encoder micro-benchmark
set -e
VARIANTS=${VARIANTS:-"head dpp"}
for rep in 1 2; do for v in $VARIANTS; do for p in patches natural; do
  echo -n "rep $rep $v $p: "; GCN10_GPU_LIB=$GRAFT_REPO_ROOT/variants/libgcn10_gpu_$v.so python tools/bench_fused.py --pattern $p --diags 0 --reps 8 | cut -c40-90
done; done; done

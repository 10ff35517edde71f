#!/bin/bash
# same-box A/B of two builds of libgcn10_gpu.so on the fused encoder micro-benchmark
set -e
for rep in 1 2; do for v in head dpp; do for p in patches natural; do
  echo -n "rep $rep $v $p: "; GCN10_GPU_LIB=$GRAFT_REPO_ROOT/variants/libgcn10_gpu_$v.so python tools/bench_fused.py --pattern $p --diags 0 --reps 8 | cut -c40-90
done; done; done

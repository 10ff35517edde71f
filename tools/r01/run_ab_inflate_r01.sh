#!/bin/bash
# same-box A/B of two builds of libgcn10_gpu.so on the inflate micro-benchmark
set -e
for rep in 1 2; do for v in head dpp; do for p in natural iid; do
  echo -n "rep $rep $v $p: "; GCN10_GPU_LIB=$GRAFT_REPO_ROOT/variants/libgcn10_gpu_$v.so python tools/bench_inflate.py --pattern $p --reps 4 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['best_ms'], d['ok'])"
done; done; done

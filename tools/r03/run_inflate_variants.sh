#!/bin/bash
# the landcover decoder of this tree ("head") against the library in variants/<name>/, same box, alternating:
# one block's 1 296 tiles (inflate + untile), best of 4
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_inflate
mkdir -p $O
for rep in 1 2; do for v in $VARIANTS head; do for p in patches natural iid; do
  if [ $v = head ]; then unset GCN10_GPU_LIB; else export GCN10_GPU_LIB=$R/variants/$v/libgcn10_gpu.so; fi
  echo -n "rep $rep $v $p: "; python3 $R/tools/bench_inflate.py --pattern $p --reps 4 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['best_ms'], d.get('ok'))"
done; done; done 2>&1 | tee $O/${1:-variants}.txt

#!/bin/bash
# stream placement: pass B' as a launch of its own (variants/bprime: the library of the commit before)
# against placement inside pass F-C, same box, alternating, 3 rounds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
: > $O/place_ab.txt
for round in 1 2 3; do
  for v in bprime inkernel; do
    for pat in natural patches; do
      if [ $v = bprime ]; then export GCN10_GPU_LIB=$R/variants/bprime/libgcn10_gpu.so; else unset GCN10_GPU_LIB; fi
      echo "$v $pat $(timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --diags 0 --reps 9 2>/dev/null)" >> $O/place_ab.txt
    done
  done
done
cat $O/place_ab.txt

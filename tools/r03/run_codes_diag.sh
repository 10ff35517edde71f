#!/bin/bash
# where pass B's time goes: the kernel leaves after phase p (option codes_stop = p + 1); whole-strip times
R=$GRAFT_REPO_ROOT
for pat in natural patches; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --diags 0 --codes-stops 0,1,2,3,4,5 --reps 5
done

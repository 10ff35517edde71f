#!/bin/bash
# where pass F-A's time goes: the kernel leaves after phase p (fused_diag 40 + p); differences of whole-strip times
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
for pat in natural patches; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --parses 1 --emits 1 --diags 0 --stats-stops 0,1,2,3,4,5 --reps 5 > $O/stats_diag_$pat.json 2>$O/stats_diag_$pat.err
  cat $O/stats_diag_$pat.json
done

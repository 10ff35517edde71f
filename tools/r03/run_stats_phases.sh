#!/bin/bash
# pass F-A (segment form): strip time with the kernel leaving after phase 0 .. 4 (option fused_stats_stop),
# noisy and patchy landcover; differences between columns = phase costs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
for pat in natural patches; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --diags 0 --stats-stops 0,1,2,3,4,5 --reps 7 | tee $O/stats_phases_$pat.json
done

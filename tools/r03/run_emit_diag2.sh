#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
for pat in natural iid; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --parses 1 --emits 1 --diags 0,8,32,16,2 --reps 4 > $O/emit_diag2_$pat.json 2>$O/emit_diag2_$pat.err
  cat $O/emit_diag2_$pat.json
done

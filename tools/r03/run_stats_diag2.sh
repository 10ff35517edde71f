#!/bin/bash
# pass F-A: without its literal histogram atomics (diag 8) and without its token write-out (diag 16); the lock-step
# emit kernel is used because it ignores those switches
R=$GRAFT_REPO_ROOT
for pat in natural patches; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --parses 1 --emits 0 --diags 0,8,16,24 --reps 5
done

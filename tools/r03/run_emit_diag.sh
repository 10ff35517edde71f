#!/bin/bash
# encoder tests, then where the wave-independent emit kernel's time goes: timing switches (invalid streams)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_deflate.py $R/tests/test_gpu_fuzz_slices.py -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt
for pat in natural iid patches; do
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --parses 1 --emits 1 --diags 0,1,4,16,20,2 --reps 4 > $O/emit_diag_$pat.json 2>$O/emit_diag_$pat.err
  cat $O/emit_diag_$pat.json
  timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --parses 1 --emits 0 --diags 0,2 --reps 4
done

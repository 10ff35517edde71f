#!/bin/bash
# pass F-A: one lane per row (rounds 1-2) against one lane per 64-px segment (round 3): encoder tests first
# (every stream must inflate to the oracle's tile), then strip-level times and arena bytes, then the
# per-kernel trace of both forms
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_deflate.py $R/tests/test_gpu_fuzz_slices.py -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt
for pat in natural patches iid; do
  for rows in 768 2304; do
    timeout -k 10 300 python3 $R/tools/bench_fused.py --pattern $pat --rows $rows --parses 0,1 --emits 0,1 --diags 0 --reps 5 > $O/fused_${pat}_$rows.json 2>$O/fused_${pat}_$rows.err
    cat $O/fused_${pat}_$rows.json
  done
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/bench_fused.py --pattern natural --rows 768 --parses 0,1 --emits 0,1 --diags 0 --reps 5 > $O/kt.log 2>&1
cp $(ls -t $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats_fused_ab_natural.csv
cut -c1-200 $O/kernel_stats_fused_ab_natural.csv | head -12
rm -rf $O/kt

#!/bin/bash
# pass F-A's own duration (kernel trace) with its timing switches: 8 = without the literal histogram atomics,
# 16 = without the token write-out, 24 = without both; lock-step emit (it ignores those switches)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for pat in natural patches; do
  echo -n "$pat:"
  for d in ${DIAGS:-0 8 16 24}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kd -- python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --emits 0 --diags $d --reps 5 > $O/kd.log 2>&1
    f=$(ls -t $O/kd/*/*kernel_stats.csv | head -1)
    echo -n " diag$d $(grep fused_stats_seg $f | awk -F, '{printf "%.1f", $6/1000}')"
    rm -rf $O/kd
  done
  echo
done | tee $O/stats_diag_trace.txt

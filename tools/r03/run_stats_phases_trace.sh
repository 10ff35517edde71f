#!/bin/bash
# pass F-A's own duration when it leaves after phase 0 .. 4 (option fused_stats_stop = p + 1), from the kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for pat in natural patches; do
  echo -n "$pat:"
  for stop in 1 2 3 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --diags 0 --stats-stops 0,$stop --reps 5 > $O/ks.log 2>&1
    f=$(ls -t $O/ks/*/*kernel_stats.csv | head -1)
    echo -n " stop$stop $(grep fused_stats_seg $f | awk -F, '{printf "%.1f", $6/1000}')"
    [ $stop = 5 ] && echo -n " full $(grep fused_stats_seg $f | awk -F, '{printf "%.1f", $7/1000}')"
    rm -rf $O/ks
  done
  echo
done | tee $O/stats_phases_trace.txt

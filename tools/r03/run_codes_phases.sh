#!/bin/bash
# pass B alone, leaving after phase p (option codes_stop = p + 1): the kernel's own duration from the trace
# (whole-strip times mislead here: a pass B that leaves early also leaves the placement's chunk totals at zero,
# and pass F-C then writes its streams on top of each other, which is faster)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for pat in natural patches; do
  echo -n "$pat:"
  for stop in 1 2 3 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kc -- python3 $R/tools/bench_fused.py --pattern $pat --rows 768 --diags 0 --codes-stops 0,$stop --reps 5 > $O/kc.log 2>&1
    f=$(ls -t $O/kc/*/*kernel_stats.csv | head -1)
    echo -n " stop$stop $(grep codes_wave $f | awk -F, '{printf "%.1f", $6/1000}')"
    [ $stop = 5 ] && echo -n " full $(grep codes_wave $f | awk -F, '{printf "%.1f", $7/1000}')"
    rm -rf $O/kc
  done
  echo
done | tee $O/codes_phases.txt

#!/bin/bash
# per-kernel durations of the fused tile encoder alone (one 768-row strip of a 36000-px block, default forms);
# with variants/<name>/libgcn10_gpu.so present and VARIANTS="name ..." also those builds, same box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_fused
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in head $VARIANTS; do
  if [ $v = head ]; then unset GCN10_GPU_LIB; else export GCN10_GPU_LIB=$R/variants/$v/libgcn10_gpu.so; fi
  for pat in natural patches; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$pat -- python3 $R/tools/bench_fused.py --pattern $pat --rows ${ROWS:-768} --diags 0 --reps 9 > $O/kt_$pat.log 2>&1
    cp $(ls -t $O/kt_$pat/*/*kernel_stats.csv | head -1) $O/kernel_stats_fused_${v}_$pat.csv
    rm -rf $O/kt_$pat
  done
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/kernel_stats_fused_*_*.csv")):
    print(f.split("/")[-1])
    for r in list(csv.DictReader(open(f)))[:5]:
        if int(r["Calls"]) >= 10:
            print("  %-45s calls %3s avg %7.1f min %7.1f us" % (r["Name"][21:66], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["MinNs"]) / 1e3))
PY

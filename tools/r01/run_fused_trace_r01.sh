#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in natural patches; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fused_trace_$p -- python3 $R/tools/bench_fused.py --pattern $p --diags 0 --reps 5 > $R/gpurun_out/fused_trace_$p.log 2>&1
python3 - $R/gpurun_out/fused_trace_$p <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fused" in r["Name"] or "codes" in r["Name"]:
            print(sys.argv[1].split("_")[-1], r["Name"][:50].ljust(50), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
done

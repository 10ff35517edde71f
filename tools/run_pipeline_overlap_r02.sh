#!/bin/bash
# kernel timeline of bin/gcn10 on noisy landcover, null sink: do the kernels of the two workers of a GPU overlap?
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r02
python3 $R/tools/bench_pipeline.py --pattern natural --blocks 8 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_ov > $R/gpurun_out/r02/overlap_plain.json
cd /tmp/gcn10_ov
rm -rf logs cn_rasters_drained cn_rasters_undrained
GCN10_SINK=null rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02/overlap_trace -- $R/bin/gcn10 -c config.txt -o > $R/gpurun_out/r02/overlap.log 2>&1
grep -h "timing" logs/rank_0.log | tail -2 | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$R/gpurun_out/r02/overlap_trace/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id"), r.get("Stream_Id")) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
tot = sum(e[1] - e[0] for e in ev)
# union of busy intervals
busy = 0; cs, ce = ev[0][0], ev[0][1]
for s, e, *_ in ev[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("dispatches %d  wall %.1f ms  sum of kernel durations %.1f ms  union (GPU busy with at least one kernel) %.1f ms" % (len(ev), (t1 - t0) / 1e6, tot / 1e6, busy / 1e6))
by = collections.Counter()
for s, e, n, q, st in ev:
    by[(q, st)] += e - s
print("by (queue, stream):", {k: round(v / 1e6, 1) for k, v in by.items()})
# how much of inflate_kernel's time overlaps other kernels
inf = [(s, e) for s, e, n, *_ in ev if "inflate_kernel" in n]
oth = [(s, e) for s, e, n, *_ in ev if "inflate_kernel" not in n]
ov = 0
for s, e in inf:
    for s2, e2 in oth:
        if s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
print("inflate_kernel: %d launches, %.1f ms in total, of which %.1f ms with another kernel running" % (len(inf), sum(e - s for s, e in inf) / 1e6, ov / 1e6))
PY
cd /tmp; rm -rf /tmp/gcn10_ov

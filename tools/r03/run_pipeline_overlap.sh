#!/bin/bash
# kernel timeline of bin/gcn10 (round 3: two-stage workers) on noisy landcover, null sink: do the kernels of the two workers of a GPU overlap?
set -e
PATTERN=${1:-natural}
SINK=${2:-null}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03_overlap
python3 $R/tools/bench_pipeline.py --pattern $PATTERN --blocks 8 --repeat 6 --modes $SINK --keep --esa-compression 8 --workdir /tmp/gcn10_ov > $R/gpurun_out/r03_overlap/overlap_plain.json
cd /tmp/gcn10_ov
rm -rf logs cn_rasters_drained cn_rasters_undrained
GCN10_SINK=$SINK rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_overlap/overlap_trace -- $R/bin/gcn10 -c config.txt -o > $R/gpurun_out/r03_overlap/overlap.log 2>&1
grep -h "timing" logs/rank_0.log | tail -2 | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$R/gpurun_out/r03_overlap/overlap_trace/**/*kernel_trace.csv", recursive=True))
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
print("$PATTERN / $SINK sink")
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
def short(n):
    for k in ("inflate_kernel", "untile_kernel", "fused_stats", "fused_emit", "deflate_codes", "expand_x_codes", "copyBuffer"):
        if k in n:
            return k
    return n[:30]
streams = collections.Counter(st for *_, st in ev)
for st, cnt in streams.most_common(4):
    e = [x for x in ev if x[4] == st]
    gaps = collections.Counter(); gapn = collections.Counter()
    for a, b in zip(e, e[1:]):
        key = (short(a[2]), short(b[2]))
        gaps[key] += max(b[0] - a[1], 0); gapn[key] += 1
    print("stream %s: %d kernels, busy %.1f ms of a span of %.1f ms; largest gaps:" % (st, len(e), sum(x[1] - x[0] for x in e) / 1e6, (e[-1][1] - e[0][0]) / 1e6))
    for k, v in gaps.most_common(4):
        print("    after %-15s before %-15s %.1f ms over %d (avg %.0f us)" % (k[0], k[1], v / 1e6, gapn[k], v / gapn[k] / 1e3))
print("inflate_kernel: %d launches, %.1f ms in total, of which %.1f ms with another kernel running" % (len(inf), sum(e - s for s, e in inf) / 1e6, ov / 1e6))
PY
cd /tmp; rm -rf /tmp/gcn10_ov

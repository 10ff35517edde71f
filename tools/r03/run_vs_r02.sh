#!/bin/bash
# Same box, same worlds: the round-2 program (variants/r02, built from commit 764dae8) against this round's.
# Steady state = (wall of N2 blocks - wall of N1 blocks) / (N2 - N1): works for both builds (round 2 has no
# "after its first block" line); this round's own steady-state line is printed beside it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
run() {  # label bin pattern repeat mode
  GCN10_BIN=$2 python3 $R/tools/bench_pipeline.py --pattern $3 --blocks 8 --repeat $4 --modes $5 --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_vs_$3 | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']['$5']
print('$1 $3 $5 blocks %d wall %.3f after_first %s rc %d' % (d['blocks_done'], d['seconds'], d['after_first_block_seconds_per_block'], d['rc']))"
}
for pat in natural patches; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 1 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_vs_$pat > /dev/null 2>&1
  for rep in 1 2; do
    for mode in null files; do
      [ $pat = natural ] && [ $mode = files ] && reps="1 2" || reps="2 6"
      for n in $reps; do
        run r02 $R/variants/r02/bin/gcn10 $pat $n $mode
        run r03 $R/bin/gcn10 $pat $n $mode
      done
    done
  done
  rm -rf /tmp/gcn10_vs_$pat
done 2>&1 | tee $O/vs_r02.txt
python3 - $O/vs_r02.txt <<'PY'
import sys, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 8 and p[3] == "blocks":
        rows[(p[0], p[1], p[2])].append((int(p[4]), float(p[6]), p[8]))
for k, v in sorted(rows.items()):
    by = collections.defaultdict(list)
    for n, w, af in v:
        by[n].append(w)
    ns = sorted(by)
    if len(ns) == 2:
        a, b = ns
        print("%s %s %s: steady state %.4f s per block (best walls: %d blocks %.3f s, %d blocks %.3f s); own line: %s" % (
            k[0], k[1], k[2], (min(by[b]) - min(by[a])) / (b - a), a, min(by[a]), b, min(by[b]), [x[2] for x in v if x[0] == b]))
PY

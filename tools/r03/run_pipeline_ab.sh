#!/bin/bash
# Round 3: the two-stage pipeline (input thread one block ahead) against input and encode in turn, buffered
# against O_DIRECT file writes, DEFLATE and raw landcover; bin/gcn10 itself, 36000^2 blocks, one GPU.
# usage: run_pipeline_ab.sh [blocks=8] [repeat=2]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
NB=${1:-8}
REP=${2:-2}
df -hT /tmp | tail -1
one() {   # tag pattern compression modes [env...]
  tag=$1; pat=$2; comp=$3; modes=$4; shift 4
  env "$@" python3 $R/tools/bench_pipeline.py --pattern $pat --blocks $NB --repeat $REP --modes $modes --keep --reuse --esa-compression $comp --workdir /tmp/gcn10_ab_${pat}_$comp > $O/$tag.json 2>$O/$tag.err
  python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for m, v in d["modes"].items():
    print("%-34s %-5s rc %d  s/block %.4f  after start-up %.4f  | %s" % (sys.argv[2], m, v["rc"], v["seconds_per_block"] or -1, v["steady_seconds_per_block"] or -1, (v["worker_seconds"] or "")[:230]))
PY
}
for pat in natural patches; do
  one ${pat}_deflate_prefetch1 $pat 8 null,files GCN10_PREFETCH_BLOCKS=1
  one ${pat}_deflate_prefetch0 $pat 8 null,files GCN10_PREFETCH_BLOCKS=0
  one ${pat}_deflate_prefetch1_direct $pat 8 files GCN10_DIRECT_IO=1
  one ${pat}_deflate_prefetch1_again $pat 8 null,files GCN10_PREFETCH_BLOCKS=1
  rm -rf /tmp/gcn10_ab_${pat}_8
done
one natural_raw_prefetch1 natural 1 null,files GCN10_PREFETCH_BLOCKS=1
one natural_raw_prefetch0 natural 1 null GCN10_PREFETCH_BLOCKS=0
rm -rf /tmp/gcn10_ab_natural_1

#!/bin/bash
# how far behind the submitted strip the worker drains (GCN10_DRAIN_LAG) x strip buffer sets; null and files
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in natural patches; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 2 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_dl_$pat > /dev/null 2>&1
  for rep in 1 2; do for cfg in "1 4" "2 4" "3 4" "2 6" "3 6" "4 6" "5 8"; do
    set -- $cfg
    echo -n "$pat drain_lag $1 buffers $2 rep $rep: "
    GCN10_DRAIN_LAG=$1 GCN10_STRIP_BUFFERS=$2 python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 2 --modes null,files --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_dl_$pat | python3 -c "import json,sys; d=json.load(sys.stdin)['modes']; print('null', d['null']['steady_seconds_per_block'], d['null']['after_first_block_seconds_per_block'], 'files', d['files']['steady_seconds_per_block'], d['files']['after_first_block_seconds_per_block'])"
  done; done
  rm -rf /tmp/gcn10_dl_$pat
done 2>&1 | tee $O/drain_lag_sweep.txt

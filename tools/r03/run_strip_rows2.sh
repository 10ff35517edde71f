#!/bin/bash
# small strips and more workers per GPU with the round-3 kernels (null sink and files), 16 blocks
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
for pat in natural patches; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 2 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_sr_$pat > /dev/null 2>&1
  for rep in 1 2; do for cfg in "256 2" "512 2" "768 2" "768 3" "768 4" "512 3" "512 4" "1024 4"; do
    set -- $cfg
    echo -n "$pat strip_rows $1 workers $2 rep $rep: "
    python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 2 --modes null,files --keep --reuse --strip-rows $1 --workers-per-gpu $2 --esa-compression 8 --workdir /tmp/gcn10_sr_$pat | python3 -c "import json,sys; d=json.load(sys.stdin)['modes']; print('null', d['null']['steady_seconds_per_block'], d['null']['after_first_block_seconds_per_block'], 'files', d['files']['steady_seconds_per_block'], d['files']['after_first_block_seconds_per_block'])"
  done; done
  rm -rf /tmp/gcn10_sr_$pat
done 2>&1 | tee $O/strip_rows_workers_sweep.txt

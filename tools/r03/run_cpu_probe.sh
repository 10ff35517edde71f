#!/bin/bash
# where the host's CPU time goes with the file sink on patchy blocks: the program's own timing lines,
# then the same run under strace -c if the box has it
R=$GRAFT_REPO_ROOT
W=/tmp/gcn10_c
python3 $R/tools/bench_pipeline.py --pattern patches --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir $W > /dev/null 2>&1
cd $W
for io in 0; do
  sed -i "s/^io_threads=.*/io_threads=$io/" config.txt
  rm -rf logs cn_rasters_drained cn_rasters_undrained
  for sink in files null; do
    rm -rf logs cn_rasters_drained cn_rasters_undrained
    if [ $sink = null ]; then export GCN10_SINK=null; else unset GCN10_SINK; fi
    $R/bin/gcn10 -c config.txt -o --gpus 1 > /dev/null 2>&1
    echo "io_threads=$io sink=$sink"
    grep -h "timing: steady\|host cpu\|pool cpu\|chunk cache" logs/rank_0.log | sed 's/^[^]]*\] //' | cut -c1-420
  done
done
which strace perf ltrace 2>/dev/null
rm -rf $W

#!/bin/bash
# kernel + memory-copy trace of bin/gcn10 itself: DEFLATE natural-pattern landcover, GPU inflate + fused encoder
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 $R/tools/bench_pipeline.py --pattern natural --blocks 6 --modes files --keep --esa-compression 8 --workdir /tmp/gcn10_pb5 > $R/gpurun_out/prof4_plain.json
cut -c100-560 $R/gpurun_out/prof4_plain.json
cd /tmp/gcn10_pb5
rm -rf logs cn_rasters_drained cn_rasters_undrained
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe4 -- $R/bin/gcn10 -c config.txt -o > $R/gpurun_out/prof4.log 2>&1
grep -h "timing" logs/rank_0.log | tail -1
for f in $R/gpurun_out/prof_pipe4/*/*kernel_stats.csv $R/gpurun_out/prof_pipe4/*/*memory_copy_stats.csv; do cut -c1-150 $f | head -12; done

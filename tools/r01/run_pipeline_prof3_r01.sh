#!/bin/bash
# kernel + memory-copy trace of bin/gcn10 itself (fused mode, 6 full-size blocks)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 $R/tools/bench_pipeline.py --blocks 6 --modes files --keep --gpu-deflate 2 --workdir /tmp/gcn10_pb4 > $R/gpurun_out/prof3_plain.json
cat $R/gpurun_out/prof3_plain.json | cut -c150-520
cd /tmp/gcn10_pb4
rm -rf logs cn_rasters_drained cn_rasters_undrained
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe3 -- $R/bin/gcn10 -c config.txt -o > $R/gpurun_out/prof3.log 2>&1
grep timing logs/rank_0.log; grep "worker seconds" logs/rank_0.log
for f in $R/gpurun_out/prof_pipe3/*/*kernel_stats.csv $R/gpurun_out/prof_pipe3/*/*memory_copy_stats.csv; do echo $f; cut -c1-150 $f; done

#!/bin/bash
# kernel + memory-copy trace of bin/gcn10 itself, round-2 configuration (DEFLATE landcover,
# GPU inflate, fused encoder), patchy and noisy landcover, 6 full-size blocks each
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for p in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $p --blocks 6 --modes files --keep --esa-compression 8 --workdir /tmp/gcn10_pf_$p > $R/gpurun_out/proff_plain_$p.json
  cd /tmp/gcn10_pf_$p
  rm -rf logs cn_rasters_drained cn_rasters_undrained
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/proff_$p -- $R/bin/gcn10 -c config.txt -o > $R/gpurun_out/proff_$p.log 2>&1
  grep -h "timing" logs/rank_0.log | tail -1 | cut -c1-160
  cp $(ls -t $R/gpurun_out/proff_$p/*/*kernel_stats.csv | head -1) $R/gpurun_out/r02_kernel_stats_cli_$p.csv
  cp $(ls -t $R/gpurun_out/proff_$p/*/*memory_copy_stats.csv | head -1) $R/gpurun_out/r02_memcopy_stats_cli_$p.csv
  cd /tmp; rm -rf /tmp/gcn10_pf_$p
done

#!/bin/bash
# host CPU seconds of bin/gcn10 against its wall time: is the 16-CPU quota what bounds the block rate?
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r02
O=$R/gpurun_out/r02/pipeline_host_cpu.txt
: > $O
for spec in "patches 32" "natural 16"; do
  set -- $spec
  python3 $R/tools/bench_pipeline.py --pattern $1 --blocks $2 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_cpu > /dev/null
  cd /tmp/gcn10_cpu
  for sink in files null; do
    rm -rf logs cn_rasters_drained cn_rasters_undrained
    if [ $sink = null ]; then GCN10_SINK=null $R/bin/gcn10 -c config.txt -o > /dev/null 2>&1; else $R/bin/gcn10 -c config.txt -o > /dev/null 2>&1; fi
    echo "== $1, $sink sink, $2 blocks" >> $O
    grep -h "timing" logs/rank_0.log | grep -v "timing gpu" | tail -2 | cut -c1-420 >> $O
  done
  cd /tmp; rm -rf /tmp/gcn10_cpu
done
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" >> $O
cat $O

#!/bin/bash
# Round-3 baseline of the product pipeline (bin/gcn10 itself: DEFLATE landcover in, GPU inflate,
# fused encoder, 18 DEFLATE GeoTIFFs out), before any round-3 change:
#   1. plain run, 8 noisy ("natural") blocks, null sink and files
#   2. rocprofv3 --kernel-trace --stats, 6 noisy blocks, null sink
#   3. three SQ counter passes (--pmc only, kernel dispatches serialised), 2 noisy blocks, null sink
#   4. the same kernel trace on patchy landcover
# Summaries land in gpurun_out/r03_baseline/; tools/r03/summarize_pmc.py turns the counter CSVs into
# per-kernel rows.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_${1:-baseline}
mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-baseline}
python3 $R/tools/bench_pipeline.py --pattern natural --blocks 8 --modes null,files --keep --esa-compression 8 --workdir /tmp/gcn10_b_nat > $O/${TAG}_plain_natural.json
cat $O/${TAG}_plain_natural.json | cut -c1-1200
cd /tmp/gcn10_b_nat
printf '1 2 3 4 5 6\n' > six.txt
printf '1 2\n' > two.txt
export GCN10_SINK=null
rm -rf logs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_natural -- $R/bin/gcn10 -c config.txt -o -l six.txt > $O/kt_natural.log 2>&1
grep -h "timing" logs/rank_0.log | cut -c1-400
cp $(ls -t $O/kt_natural/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_cli_natural.csv
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_INT32 SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf logs
  rocprofv3 --pmc $set --output-format csv -d $O/pmc${i}_natural -- $R/bin/gcn10 -c config.txt -o -l two.txt > $O/pmc${i}_natural.log 2>&1 || echo "pmc pass $i failed"
done
unset GCN10_SINK
cd /tmp; rm -rf /tmp/gcn10_b_nat
python3 $R/tools/bench_pipeline.py --pattern patches --blocks 8 --modes null,files --keep --esa-compression 8 --workdir /tmp/gcn10_b_pat > $O/${TAG}_plain_patches.json
cat $O/${TAG}_plain_patches.json | cut -c1-1200
cd /tmp/gcn10_b_pat
printf '1 2 3 4 5 6\n' > six.txt
export GCN10_SINK=null
rm -rf logs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_patches -- $R/bin/gcn10 -c config.txt -o -l six.txt > $O/kt_patches.log 2>&1
cp $(ls -t $O/kt_patches/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_cli_patches.csv
cd /tmp; rm -rf /tmp/gcn10_b_pat
python3 $R/tools/r03/summarize_pmc.py $O > $O/${TAG}_pmc_summary.txt
cat $O/${TAG}_pmc_summary.txt
# keep only the summaries (the raw traces are large)
rm -rf $O/kt_natural $O/kt_patches
for i in 1 2 3; do
  f=$(ls $O/pmc${i}_natural/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && rm -rf $O/pmc${i}_natural
done

#!/bin/bash
# Dress rehearsal of an 8-GPU node on the one GPU of the box, for the PRODUCT (bin/gcn10), not the bench:
# GCN10_REHEARSE_GPUS=8 -> 8 logical GPUs x 2 workers = 16 block workers (+16 input threads), one I/O pool,
# eight "timing gpu N" lines, every block id exactly once, exit code 0; then the same with one worker made to
# fail (GCN10_TEST_FAIL_BLOCK: an MPI_Abort-class error at one block): exit code 1.
# 64 blocks of 3000^2 px (file sink) and 16 blocks of 36000^2 (null sink: 16 workers' buffers at full size).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_rehearsal
mkdir -p $O
export TMPDIR=/tmp
export GCN10_REHEARSE_GPUS=8
python3 $R/tools/bench_pipeline.py --size 3000 --pattern natural --blocks 8 --repeat 8 --gpus 8 --modes files --keep --esa-compression 8 --workdir /tmp/gcn10_reh > $O/rehearse_64_small_blocks.json 2>$O/rehearse.err
cd /tmp/gcn10_reh
rm -rf logs cn_rasters_drained cn_rasters_undrained
$R/bin/gcn10 -c config.txt -o --gpus 8 > $O/run.out 2>&1
echo "exit code $?" | tee $O/rehearse_summary.txt
{
  echo "log files: $(ls logs | wc -l)"
  echo "processing lines: $(cat logs/rank_*.log | grep -c 'processing block')  distinct ids: $(cat logs/rank_*.log | grep -o 'processing block [0-9]*' | sort -u | wc -l)"
  echo "rasters written: $(ls cn_rasters_drained cn_rasters_undrained | grep -c tif)  (64 blocks x 18)"
  grep -h "starting processing\|processed .* blocks on" logs/rank_0.log | cut -c1-200
  grep -h "timing" logs/rank_0.log | cut -c1-420
} | tee -a $O/rehearse_summary.txt
# one worker meets an MPI_Abort-class error at block 37 (test hook): the run stops and exits with code 1
rm -rf logs cn_rasters_drained cn_rasters_undrained
GCN10_TEST_FAIL_BLOCK=37 $R/bin/gcn10 -c config.txt -o --gpus 8 > $O/run_fail.out 2>&1
echo "one worker fails at block 37: exit code $? ; blocks started: $(cat logs/rank_*.log | grep -c 'processing block') of 64; $(grep -h 'malloc failed for block' logs/rank_*.log | cut -c1-120)" | tee -a $O/rehearse_summary.txt
cd /tmp; rm -rf /tmp/gcn10_reh
# full-size blocks, 16 workers' worth of buffers on one card, null sink
python3 $R/tools/bench_pipeline.py --pattern patches --blocks 8 --repeat 4 --gpus 8 --modes null --esa-compression 8 --workdir /tmp/gcn10_reh2 > $O/rehearse_32_full_size_blocks_null.json 2>>$O/rehearse.err
python3 -c "
import json
d=json.load(open('$O/rehearse_32_full_size_blocks_null.json'))['modes']['null']
print('full-size blocks, 16 workers on one card, null sink: rc', d['rc'], 'blocks', d['blocks_done'], 'wall', d['seconds'], 's =', round(d['seconds'] / max(d['blocks_done'], 1), 4), 's per block incl. 16 workers\' start-up;', (d['worker_seconds'] or '')[:160])" | tee -a $O/rehearse_summary.txt

R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/tools/bench_pipeline.py --blocks 12 --modes files --workdir /tmp/gcn10_pb3 > $R/gpurun_out/prof_pipe.log 2>&1 || true
grep -h '^{"size"' $R/gpurun_out/prof_pipe.log | cut -c180-520
f=$(ls -t $R/gpurun_out/prof_pipe/*/*kernel_stats.csv | head -1)
cut -c1-140 $f

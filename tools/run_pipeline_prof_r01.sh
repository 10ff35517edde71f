set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
python tools/bench_pipeline.py --blocks 3 > gpurun_out/pipeline_gd2.json 2> gpurun_out/pipeline_gd2.err
cat gpurun_out/pipeline_gd2.json
# deflate kernel time under rocprof: a small standalone run through the test
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_defl -- python3 $R/tools/bench_pipeline.py --blocks 1 --modes files --workdir /tmp/gcn10_pb2 > $R/gpurun_out/prof_defl.log 2>&1 || true
cat $R/gpurun_out/prof_defl/*/*kernel_stats.csv | cut -c1-200

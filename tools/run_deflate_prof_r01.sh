R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_deflate.py tests/test_cli.py -m gpu -x -q -s > gpurun_out/pytest_deflate2.log 2>&1; echo pytest rc=$?
grep -E "gpu .* zlib|passed|failed|Error" gpurun_out/pytest_deflate2.log | head -20
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_defl2 -- python3 $R/tools/bench_pipeline.py --blocks 2 --modes files --workdir /tmp/gcn10_pb2 > $R/gpurun_out/prof_defl2.log 2>&1 || true
tail -1 $R/gpurun_out/prof_defl2.log | cut -c1-700
cat $R/gpurun_out/prof_defl2/*/*kernel_stats.csv | cut -c1-160

# parity tests of the strip kernels, the launch-shape sweep and the bench line, one box.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/parity_r02.log 2>&1 || { tail -30 gpurun_out/parity_r02.log; exit 1; }
tail -3 gpurun_out/parity_r02.log
timeout -k 10 600 python3 tools/tune_strip.py > gpurun_out/tune_r02.jsonl 2> gpurun_out/tune_r02.err
echo "tune done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r02_new.json 2>&1
echo "bench done"

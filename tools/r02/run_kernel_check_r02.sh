# parity tests of the strip kernels, then the bench line twice (with / without the placement calibration), one box.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/parity_r02.log 2>&1 || { tail -30 gpurun_out/parity_r02.log; exit 1; }
tail -3 gpurun_out/parity_r02.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_tuned.json 2> gpurun_out/bench_r02_tuned.err
echo "bench tuned done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-tune --no-cpu-baseline > gpurun_out/bench_r02_untuned.json 2>&1
echo "bench untuned done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload config4 > gpurun_out/bench_r02_config4.json 2>&1
echo "bench config4 done"

# re-validation of the two GPU codecs after the fence change of commit 1dbdfde, fixed budgets, seeds recorded
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 tests/fuzz_fused_encoder.py --cases 2000 --seed 20261004 > gpurun_out/fuzz_fused_r02.log 2>&1
tail -1 gpurun_out/fuzz_fused_r02.log
timeout -k 10 600 python3 tools/fuzz_inflate.py --streams 2048 --seed 20261004 > gpurun_out/fuzz_inflate_r02.log 2>&1
tail -1 gpurun_out/fuzz_inflate_r02.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz_slices.py -x -q > gpurun_out/fuzz_slices_r02.log 2>&1 || { tail -20 gpurun_out/fuzz_slices_r02.log; exit 1; }
tail -2 gpurun_out/fuzz_slices_r02.log

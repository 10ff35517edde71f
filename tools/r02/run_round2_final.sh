# final checks of the round on one box: full GPU suite, smoke, profile recipe
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests_r02.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests_r02.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests_r02.log
python3 __graft_entry__.py smoke 2>&1 | tail -1
bash $R/profiles/run_profiles_r02.sh

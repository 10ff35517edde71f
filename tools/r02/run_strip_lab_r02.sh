# same-box A/B of the single-raster strip kernel variants (tools/strip_lab.hip, interleaved rounds)
# next to copy kernels, then the product library's own variants (tools/tune_strip.py) on the same box.
# Usage: gpurun -- bash tools/r02/run_strip_lab_r02.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/strip_lab 36001 3000 1 2 check > $R/gpurun_out/strip_lab_36001.jsonl 2>&1
echo "36001 done"
timeout -k 10 400 $R/tools/strip_lab 36000 36000 6 6 main > $R/gpurun_out/strip_lab.jsonl 2>&1
echo "lab done"
cd $R
timeout -k 10 600 python3 tools/tune_strip.py > gpurun_out/tune_r02.jsonl 2> gpurun_out/tune_r02.err
echo "tune done"

# same-box A/B of the single-raster strip kernel variants (tools/strip_lab.hip) next to copy kernels,
# then the product bench line on the same box.  Usage: gpurun -- bash tools/run_strip_lab_r02.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/strip_lab 36001 3000 5 quick > $R/gpurun_out/strip_lab_36001.jsonl 2>&1
echo "36001 done"
timeout -k 10 400 $R/tools/strip_lab 36000 36000 20 > $R/gpurun_out/strip_lab.jsonl 2>&1
echo "lab done"
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/bench_r02_base.json 2>&1
echo "bench done"

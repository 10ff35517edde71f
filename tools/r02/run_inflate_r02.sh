# inflate kernel after the 16 KiB window change: unit tests, fuzz, kernel bench, pipeline
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_inflate.py tests/test_gpu_fuzz_slices.py -x -q 2>&1 | tail -2
timeout -k 10 600 python3 tools/fuzz_inflate.py --streams 4096 --seed 31337 | tail -1
for p in patches natural iid; do timeout -k 10 300 python3 tools/bench_inflate.py --pattern $p > gpurun_out/r02_inflate_$p.json 2>&1; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r02_inflate_$p.json') if l.startswith('{')][-1]); print('$p', {k:d[k] for k in d if k in ('gpu_ms','gpu_GBps','ratio','host_core_ms','speedup_vs_one_core','ms','GBps')})"; done
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes null --esa-compression 8 > gpurun_out/r02_pipeline_natural_null.json
python3 -c "
import json; d=json.load(open('gpurun_out/r02_pipeline_natural_null.json'))
for k,m in d['modes'].items(): print('natural', k, m['seconds'], m['seconds_per_block'], '| after start-up:', m['seconds_after_startup'], m['steady_seconds_per_block'])"

# A/B of the inflate dispatch order (file order vs longest stream first) through the whole program, one box
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes null --esa-compression 8 --keep --workdir /tmp/gcn10_ab > gpurun_out/inflate_order_lpt.json
GCN10_INFLATE_FILE_ORDER=1 timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes null --esa-compression 8 --reuse --keep --workdir /tmp/gcn10_ab > gpurun_out/inflate_order_file.json
timeout -k 10 900 python tools/bench_pipeline.py --pattern natural --blocks 12 --modes null --esa-compression 8 --reuse --workdir /tmp/gcn10_ab > gpurun_out/inflate_order_lpt2.json
for f in inflate_order_lpt inflate_order_file inflate_order_lpt2; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json'))
for k,m in d['modes'].items(): print('$f', k, m['seconds'], m['seconds_per_block'], '| after start-up:', m['seconds_after_startup'], m['steady_seconds_per_block'])"; done

set -e
cd $GRAFT_REPO_ROOT
for pat in patches natural; do
python tools/bench_pipeline.py --pattern $pat --blocks 16 --modes null --esa-compression 8 --keep --workdir /tmp/gcn10_sr > gpurun_out/sr_${pat}_1024.json
for sr in 768 1280 512; do
python tools/bench_pipeline.py --pattern $pat --blocks 16 --modes null --esa-compression 8 --reuse --keep --workdir /tmp/gcn10_sr --strip-rows $sr > gpurun_out/sr_${pat}_$sr.json
done
python tools/bench_pipeline.py --pattern $pat --blocks 16 --modes null --esa-compression 8 --reuse --workdir /tmp/gcn10_sr > gpurun_out/sr_${pat}_1024b.json
for sr in 1024 768 1280 512 1024b; do python3 -c "
import json; d=json.load(open('gpurun_out/sr_${pat}_$sr.json'))
for k,m in d['modes'].items(): print('$pat', '$sr', m['seconds_per_block'], '| after start-up:', m['steady_seconds_per_block'])"; done
done

#!/bin/bash
# CLI tests, then files against null sink, 48 patchy blocks and 16 noisy ones (steady state after each worker's first block)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 800 python3 -m pytest $R/tests/test_cli.py -m gpu -x -q > $O/cli_tests.txt 2>&1; tail -3 $O/cli_tests.txt
show() { python3 -c "
import json
d=json.load(open('$1'))
for m,v in d['modes'].items():
    print('%-28s %-5s rc %d blocks %3d  s/block after first %s  cpu-s/block %s | %s' % ('$2', m, v['rc'], v['blocks_done'], v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block'], (v['worker_seconds'] or '')[:260]))"; }
for rep in 1 2; do
python3 $R/tools/bench_pipeline.py --pattern patches --blocks 8 --repeat 6 --modes null,files --esa-compression 8 --workdir /tmp/gcn10_fc > $O/files_check_patches_$rep.json 2>/dev/null; show $O/files_check_patches_$rep.json patches
done
python3 $R/tools/bench_pipeline.py --pattern natural --blocks 8 --repeat 2 --modes null,files --esa-compression 8 --workdir /tmp/gcn10_fc > $O/files_check_natural.json 2>/dev/null; show $O/files_check_natural.json natural

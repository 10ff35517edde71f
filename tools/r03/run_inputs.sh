#!/bin/bash
# Landcover input kinds through the round-3 input stage, full-size blocks, steady state after every worker's
# first block: DEFLATE, DEFLATE + predictor 2, uncompressed (the north star's literal path: raw bytes -> pinned
# ring -> hipMemcpyAsync -> untile on the GPU), LZW (host reader); and the host's CPU seconds per block.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pipeline
mkdir -p $O
export TMPDIR=/tmp
show() { python3 -c "
import json,sys
d=json.load(open('$1'))
for m,v in d['modes'].items():
    print('%-44s %-5s rc %d blocks %3d  s/block after first %s  (incl. start-up %.4f)  cpu-s/block %s  pinned %s MB | %s' % ('$2', m, v['rc'], v['blocks_done'], v['after_first_block_seconds_per_block'], v['seconds_per_block'] or -1, v['host_cpu_seconds_per_block'], v['pinned_MB'], (v['worker_seconds'] or '')[:150]))"; }
one() { # tag pattern comp pred repeat modes
  python3 $R/tools/bench_pipeline.py --pattern $2 --blocks 8 --repeat $5 --modes $6 --esa-compression $3 --esa-predictor $4 --workdir /tmp/gcn10_in > $O/$1.json 2>$O/$1.err
  show $O/$1.json $1
}
if [ "$1" != "natural-only" ]; then
one input_patches_deflate patches 8 1 6 null,files
one input_patches_deflate_predictor2 patches 8 2 6 null
one input_patches_raw patches 1 1 6 null,files
fi
# (LZW landcover stays with the host reader; the pure-Python LZW writer of tests/tiffutil.py is far too slow for a
# full-size world: covered at test size by tests/test_cli.py::test_raw_and_predictor2_landcover_through_the_gpu_side)
one input_natural_raw natural 1 1 6 null
one input_natural_deflate natural 8 1 6 null
one input_natural_deflate_files natural 8 1 2 files

#!/bin/bash
# inflate with an 8 KiB LDS ring (round 3 default, ten streams per CU) against the 16 KiB ring of round 2
# (six per CU): unit tests first, then the micro-benchmark on the three patterns, then the pipeline (natural, null sink)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_inflate
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_inflate.py $R/tests/test_gpu_fuzz_slices.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for rep in 1 2; do for v in w8k w16k; do for p in patches natural iid; do
  lib=$R/gcn10_amd/libgcn10_gpu.so; [ $v = w16k ] && lib=$R/variants/libgcn10_gpu_w16k.so
  echo -n "rep $rep $v $p: "; GCN10_GPU_LIB=$lib python3 $R/tools/bench_inflate.py --pattern $p --reps 4 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['best_ms'], d['ok'])"
done; done; done 2>&1 | tee $O/inflate_window_ab.txt
python3 $R/tools/bench_pipeline.py --pattern natural --blocks 8 --repeat 2 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_iab > /dev/null 2>&1
for rep in 1 2 3; do for v in w8k w16k; do
  lib=$R/gcn10_amd/libgcn10_gpu.so; [ $v = w16k ] && lib=$R/variants/libgcn10_gpu_w16k.so
  echo -n "pipeline rep $rep $v natural null: "
  GCN10_GPU_LIB=$lib python3 $R/tools/bench_pipeline.py --pattern natural --blocks 8 --repeat 2 --modes null --keep --reuse --esa-compression 8 --workdir /tmp/gcn10_iab | python3 -c "import json,sys; d=json.load(sys.stdin)['modes']['null']; print(d['seconds_per_block'], d['steady_seconds_per_block'], d['worker_seconds'][:200])"
done; done 2>&1 | tee -a $O/inflate_window_ab.txt
rm -rf /tmp/gcn10_iab

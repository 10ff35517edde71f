R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03_pipeline
for pat in patches natural; do
  python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes null --keep --esa-compression 8 --workdir /tmp/gcn10_w3_$pat > /dev/null 2>&1
  for rep in 1 2; do for wk in 2 3 4; do
    modes=null; [ $pat = patches ] && modes=null,files
    echo -n "$pat workers $wk rep $rep (16 hardware queues): "
    python3 $R/tools/bench_pipeline.py --pattern $pat --blocks 8 --repeat 9 --modes $modes --keep --reuse --workers-per-gpu $wk --esa-compression 8 --workdir /tmp/gcn10_w3_$pat | python3 -c "
import json,sys
d=json.load(sys.stdin)['modes']
print(' '.join('%s %s (cpu %s)' % (m, v['after_first_block_seconds_per_block'], v['host_cpu_seconds_per_block']) for m, v in d.items()))"
  done; done
  rm -rf /tmp/gcn10_w3_$pat
done 2>&1 | tee $R/gpurun_out/r03_pipeline/workers_with_16_hw_queues_72_blocks.txt

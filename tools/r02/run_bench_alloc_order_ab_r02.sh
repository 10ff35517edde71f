set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do for ef in 1 0; do
GCN10_BENCH_EXTRA_FIRST=$ef timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_extra_${ef}_$i.json 2>&1
python3 -c "
import json; r=json.loads([l for l in open('gpurun_out/ab_extra_${ef}_$i.json') if l.startswith('{')][-1]); ro=r['roofline']
print('extra_first=$ef run $i', r['value'], ro['avg_launch_ms'], ro['frac'], 'copy', ro['copy_ceiling']['avg_ms'], 'also', r['also']['avg_launch_ms'], r['also']['round1_style_ms_per_launch'], ro['placement']['allocations_tried_best_ms'])"
done; done

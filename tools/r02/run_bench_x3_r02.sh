set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r02_multiarena_$i.json 2>&1
python3 -c "
import json; r=json.loads([l for l in open('gpurun_out/bench_r02_multiarena_$i.json') if l.startswith('{')][-1]); ro=r['roofline']
print($i, r['value'], r['ms_per_step'], ro['kernel'], ro['avg_launch_ms'], ro['frac'], 'copy', ro['copy_ceiling']['avg_ms'], ro['frac_of_copy'], ro['placement'])"
done

# timing experiments on pass F-A of the fused encoder (fused_diag 8: no per-literal histogram atomics, 16: no token
# write-out): per-kernel averages from rocprofv3 --kernel-trace --stats, one strip of 768 rows of a 36000-px block
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pat in natural patches; do for d in 0 8 16 24; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fsd_${pat}_$d -- python3 $R/tools/bench_fused.py --diags $d --rows 768 --pattern $pat --reps 8 > $R/gpurun_out/fsd_${pat}_$d.json 2>/dev/null
f=$(ls -t $R/gpurun_out/fsd_${pat}_$d/*/*kernel_stats.csv | head -1)
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'fused_stats' in n or 'fused_emit' in n or 'codes_wave' in n: print('$pat diag $d', n.split('(')[0].split('::')[-1][:28], 'avg %.1f us'%(float(r['AverageNs'])/1e3))"
done; done

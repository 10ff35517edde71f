cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/bench_pipeline.py --blocks 24 --modes files --workdir /tmp/pbw --keep --workers-per-gpu 2 | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); m=b['modes']['files']; print('w2', m['seconds'], m['cn_gpx_per_s'], m['worker_seconds'])"
for w in 3 4 2 1; do python tools/bench_pipeline.py --blocks 24 --modes files --workdir /tmp/pbw --keep --reuse --workers-per-gpu $w | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); m=b['modes']['files']; print('w$w', m['seconds'], m['cn_gpx_per_s'], m['worker_seconds'])"; done
rm -rf /tmp/pbw

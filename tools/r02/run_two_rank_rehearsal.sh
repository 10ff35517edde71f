cd $GRAFT_REPO_ROOT
export GCN10_DIST_BACKEND=gloo
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --size 12000 > gpurun_out/bench_2ranks_gloo.log 2>&1
echo rc=$?
tail -2 gpurun_out/bench_2ranks_gloo.log | cut -c1-900

# profile recipe + a few side measurements on the same box
set -e
R=$GRAFT_REPO_ROOT
bash $R/profiles/run_profiles_r02.sh
cd /tmp
timeout -k 10 300 $R/tools/strip_lab 36000 36000 4 5 policy > $R/gpurun_out/strip_lab_policy.jsonl 2>&1
echo policy done
cd $R
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --size 36001 > gpurun_out/bench_r02_36001.json 2>&1
echo 36001 done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload config4 > gpurun_out/bench_r02_config4.json 2>&1
echo config4 done
timeout -k 10 300 python3 bench.py --gpus 2 --oversubscribe --steps 10 --warmup 2 --no-cpu-baseline --size 20000 > gpurun_out/bench_r02_2ranks_one_gpu.json 2>&1
echo 2ranks done
timeout -k 10 300 python3 bench.py --gpus 2 --oversubscribe --steps 10 --warmup 2 --no-cpu-baseline --size 20000 --scaling strong > gpurun_out/bench_r02_2ranks_strong_one_gpu.json 2>&1
echo strong done

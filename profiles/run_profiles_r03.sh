# Round-3 profile recipe (run on the GPU box through gpurun): the default bench command under
# rocprofv3 -- kernel trace, two PMC passes (FETCH_SIZE / WRITE_SIZE) and two SQ passes -- all with
# --no-cpu-baseline (no child process under the profiler), then the plain bench line on the same box.
# bench.py calibrates placement and launch shape first (gcn10_gpu_tune_single_raster), so the trace
# holds many untimed launches of several variants: the summaries below take, per kernel, only the
# dispatches of the timed steps (the last K of the chosen variant before the first stream_copy_kernel,
# and the 20 dispatch-timed launches of the copy and of the 18-raster kernel).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
K=20
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps $K --warmup 5 --no-cpu-baseline > $O/kt.json 2> $O/kt.err
echo kt done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline > $O/write.json 2> $O/write.err
echo write done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1 -- python3 $R/bench.py --steps 4 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline > $O/sq1.json 2> $O/sq1.err
echo sq1 done
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 4 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline > $O/sq2.json 2> $O/sq2.err
echo sq2 done
python3 $R/bench.py --steps $K --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
echo bench done
python3 $R/profiles/summarize_r03.py $O $O/summary $K   # copy gpurun_out/r03/summary/* to profiles/r03/ afterwards (only gpurun_out/ comes back from the box)

set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
$R/tools/hbm_streams > $R/gpurun_out/hbm_streams.jsonl 2>&1
echo streams done
python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pattern patches > $R/gpurun_out/bench_patches.log 2>&1
echo patches done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/prof_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_sq.log 2>&1
echo sq done
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_sq2.log 2>&1
echo sq2 done

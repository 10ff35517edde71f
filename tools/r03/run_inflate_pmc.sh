#!/bin/bash
# instruction counters of the landcover decoder alone (tools/bench_inflate.py, one pattern), head and variants
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_inflate
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in $VARIANTS head; do
  if [ $v = head ]; then unset GCN10_GPU_LIB; else export GCN10_GPU_LIB=$R/variants/$v/libgcn10_gpu.so; fi
  for p in ${PATTERNS:-natural}; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc -- python3 $R/tools/bench_inflate.py --pattern $p --reps 1 > $O/pmc.log 2>&1
    f=$(ls -t $O/pmc/*/*counter_collection.csv | head -1)
    python3 - "$f" "$v $p" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "inflate_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]["v"] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n["d"] += 1
print(sys.argv[2], "dispatches", n["d"], {k: round(v["v"] / max(n["d"], 1) / 1e6, 1) for k, v in acc.items()}, "(millions per dispatch)")
PY
    rm -rf $O/pmc
  done
done 2>&1 | tee $O/${1:-pmc}.txt

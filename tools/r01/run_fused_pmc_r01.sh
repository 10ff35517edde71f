#!/bin/bash
# SQ counters of the fused encoder kernels on one natural-pattern strip
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/fused_pmc1 -- python3 $R/tools/bench_fused.py --pattern natural --diags 0 --reps 1 > $R/gpurun_out/fused_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/fused_pmc2 -- python3 $R/tools/bench_fused.py --pattern natural --diags 0 --reps 1 > $R/gpurun_out/fused_pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("fused_pmc1", "fused_pmc2"):
    for f in glob.glob(R + "/gpurun_out/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "fused" in k or "codes" in k:
                print(d, k, {a: round(b / 2) for a, b in v.items()})
PY

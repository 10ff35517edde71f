#!/bin/bash
# re-validation of the two GPU codecs after round 3's rewrites (segment-parallel parse, wave-independent bit packing,
# placed extents, register-resident merge, tile flags): fixed budgets, seeds recorded
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_fuzz
mkdir -p $O
cd $R
timeout -k 10 500 python3 tests/fuzz_fused_encoder.py --cases 3000 --seed 20261005 > $O/fuzz_fused.log 2>&1
echo "fused encoder: $(tail -1 $O/fuzz_fused.log)" | tee $O/fuzz_revalidation.txt
timeout -k 10 400 python3 tools/fuzz_inflate.py --streams 3072 --seed 20261005 > $O/fuzz_inflate.log 2>&1
echo "inflate: $(tail -1 $O/fuzz_inflate.log)" | tee -a $O/fuzz_revalidation.txt

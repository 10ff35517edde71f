#!/bin/bash
# decoder wave against copier wave: inflate_diag 1 = the copier carries nothing out, 2 = the decoder hands over empty batches
R=$GRAFT_REPO_ROOT
for p in patches natural iid; do for d in 0 1 3 4; do
  echo -n "$p diag $d: "; python3 $R/tools/bench_inflate.py --pattern $p --reps 4 --diag $d | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['best_ms'])"
done; done

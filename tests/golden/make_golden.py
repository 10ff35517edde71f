"""Regenerates tests/golden/oracle_blocks.json from the ORACLE (not the reference).

The reference cannot be built or run in this image (its sources need gdal.h) and
holds no golden rasters, so these digests only freeze the oracle's own output
on seeded synthetic blocks.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import cn_oracle_c as oc  # noqa: E402
from oracle import cn_oracle_np as onp  # noqa: E402
from tests.util import make_block  # noqa: E402

LOOKUPS = os.path.join(ROOT, "tests", "golden", "lookups")
tables = np.stack([oc.load_lookup_table(os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc)))[0]
                   for hc in onp.HCS for arc in onp.ARCS])
cases = []
for seed, H, W, hsy, hsx, nasty in [(1234, 300, 300, 12, 12, False), (1235, 257, 131, 11, 6, True),
                                    (1236, 64, 1040, 3, 42, True), (1237, 1, 17, 1, 1, True)]:
    esa, gt, coarse, sgt = make_block(seed, H, W, hsy, hsx, nasty=nasty)
    out = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    assert np.array_equal(out, onp.process_block_mem(esa, gt, coarse, sgt, tables))
    cases.append({"seed": seed, "H": H, "W": W, "hsy": hsy, "hsx": hsx, "nasty": nasty,
                  "sha256": [hashlib.sha256(out[r].tobytes()).hexdigest() for r in range(18)]})
json.dump({"generator": "tests/golden/make_golden.py (oracle/cn_oracle.c)", "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "oracle_blocks.json"), "w"), indent=1)
print("wrote", len(cases), "cases")

#!/usr/bin/env python3
"""Is a process's "slow state" (every candidate raster slow) tied to where the LANDCOVER lies?  (GPU only.)
Allocates landcover copy A, ten candidate rasters, landcover copy B, ten more candidates, landcover copy C,
and times the plain copy landcover -> candidate for every pair (interleaved, dispatch-timed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from gcn10_amd import gpu  # noqa: E402

size = 36000
npix = size * size
eng = gpu.Engine(0)
esa, _, _, _ = bench.synth_block(1, size, "iid")
srcs, cands = [], []
srcs.append(eng.upload(esa))
cands += [eng.alloc(npix) for _ in range(10)]
srcs.append(eng.upload(esa))
cands += [eng.alloc(npix) for _ in range(10)]
srcs.append(eng.upload(esa))
ev = [(eng.event_create(), eng.event_create()) for _ in range(3)]
nb = npix - npix % 16
res = np.zeros((len(srcs), len(cands)))
for rnd in range(2):
    for i, s in enumerate(srcs):
        for j, c in enumerate(cands):
            eng.stream_copy(s.ptr, c.ptr, nb)
            for k in range(3):
                eng.time_next_strip(*ev[k])
                eng.stream_copy(s.ptr, c.ptr, nb)
            eng.sync()
            res[i, j] += sorted(eng.elapsed_ms(*ev[k]) for k in range(3))[1] / 2
print(json.dumps({"src_ptrs": [hex(s.ptr) for s in srcs], "cand_ptrs": [hex(c.ptr) for c in cands],
                  "copy_ms": [[round(float(v), 4) for v in row] for row in res]}))

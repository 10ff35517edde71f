#!/usr/bin/env python3
"""Kernel time of N consecutive identical launches (dispatch-timed): is the rate steady in time?  Phases: (a) strip
launches back to back, (b) the same after a 0.5 s pause of the host, (c) right after freeing 20 GB of buffers
(the driver clears freed VRAM in the background), (d) plain copies.  (GPU only.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from gcn10_amd import gpu, host  # noqa: E402

N = int(os.environ.get("N", "300"))
size = 36000
npix = size * size
eng = gpu.Engine(0)
eng.set_tables(host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups")))
esa, gt, coarse, soil_gt = bench.synth_block(1, size, "iid")
hs = coarse.shape[0]
ci, cj = host.build_index_maps(gt, soil_gt, size, size, hs, hs)
d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
out = eng.alloc(npix)
ballast = [eng.alloc(1 << 30) for _ in range(20)]
for b in ballast:
    eng.memset(b.ptr, 1, 1 << 30)
eng.sync()
eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
ev = [(eng.event_create(), eng.event_create()) for _ in range(N)]
ptrs = [None] * 18
ptrs[1] = out.ptr


def series(fn, label, before=None):
    for _ in range(3):
        fn()
    eng.sync()
    if before:
        before()
    t0 = time.monotonic()
    for i in range(N):
        eng.time_next_strip(*ev[i])
        fn()
    eng.sync()
    wall = time.monotonic() - t0
    ms = np.array([eng.elapsed_ms(*ev[i]) for i in range(N)])
    # medians of consecutive groups of 10 launches
    groups = [round(float(np.median(ms[i:i + 10])), 4) for i in range(0, N, 10)]
    print(json.dumps({"phase": label, "launches": N, "wall_ms": round(wall * 1e3, 1), "avg_ms": round(float(ms.mean()), 4),
                      "min_ms": round(float(ms.min()), 4), "max_ms": round(float(ms.max()), 4),
                      "median_of_each_10": groups}))
    sys.stdout.flush()


def strip():
    eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 1, 2, ptrs)


def copy():
    eng.stream_copy(d_esa.ptr, out.ptr, npix - npix % 16)


def free_ballast():
    for b in ballast:
        b.close()


series(strip, "strip kernel, back to back")
series(strip, "strip kernel, again")
series(strip, "strip kernel, after a 0.5 s pause", before=lambda: time.sleep(0.5))
series(strip, "strip kernel, right after freeing 20 GiB", before=free_ballast)
series(strip, "strip kernel, once more")
series(copy, "plain copy")

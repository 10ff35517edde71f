#!/usr/bin/env python3
"""Alloc / write / check / free cycles of gcn10_gpu_malloc_spread buffers, many rounds in one process: does a range
that is released and built again ever show stale contents or lose writes?  (GPU only.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from gcn10_amd import gpu  # noqa: E402

eng = gpu.Engine(0)
rng = np.random.default_rng(7)
n = 16 * 400000          # 6.4 MB
bad = 0
seen_ptrs = {}
for rnd in range(int(os.environ.get("ROUNDS", "60"))):
    src = rng.integers(0, 256, n, dtype=np.uint8)
    a = eng.upload(src)
    recipes = [(2 << 20, 0, 1), (2 << 20, 8 << 20, 1), (4 << 20, 16 << 20, 2), (0, 64 << 20, 1)]
    bufs = [eng.alloc_spread(n + 64, *recipes[k % 4]) for k in range(4)]
    for b in bufs:
        seen_ptrs[b.ptr] = seen_ptrs.get(b.ptr, 0) + 1
        eng.memset(b.ptr, 0x77, n + 64)
    for k, b in enumerate(bufs):
        eng.stream_copy(a.ptr, b.ptr, n)
    eng.sync()
    for k, b in enumerate(bufs):
        got = eng.download(b.ptr, (n + 64,))
        if not (np.array_equal(got[:n], src) and (got[n:] == 0x77).all()):
            bad += 1
            diff = np.flatnonzero(got[:n] != src)
            print("round %d buffer %d (%#x): %d bytes differ, first at %d (chunk %d), value %#x want %#x; tail ok %s"
                  % (rnd, k, b.ptr, diff.size, diff[0] if diff.size else -1, (diff[0] >> 21) if diff.size else -1,
                     got[diff[0]] if diff.size else 0, src[diff[0]] if diff.size else 0, (got[n:] == 0x77).all()))
    for b in bufs:
        b.close()
    a.close()
print("rounds done, bad buffers: %d; distinct addresses %d, most reused %d times" %
      (bad, len(seen_ptrs), max(seen_ptrs.values())))

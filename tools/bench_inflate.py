#!/usr/bin/env python3
"""Throughput of gcn10_gpu_inflate_tiles on one block's worth of landcover tiles
(1296 zlib streams of 1024 x 1024 pixels -> 36864 x 36864 raster).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gcn10_amd import gpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pattern", default="patches")
    ap.add_argument("--distinct", type=int, default=36)
    ap.add_argument("--tiles-per-side", type=int, default=36)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--diag", type=int, default=0, help="gcn10_gpu_set_option inflate_diag (timing only, output invalid)")
    a = ap.parse_args()
    T = 1024
    side = 6 * T
    esa, _, _, _ = bench.synth_block(1, side, a.pattern)
    tiles = [np.ascontiguousarray(esa[(k // 6) * T:(k // 6 + 1) * T, (k % 6) * T:(k % 6 + 1) * T]) for k in range(min(a.distinct, 36))]
    t0 = time.time()
    streams = [zlib.compress(t.tobytes(), a.level) for t in tiles]
    t_comp = time.time() - t0
    t0 = time.time()
    for s in streams:
        zlib.decompress(s)
    host_inflate_s = (time.time() - t0) / len(streams)
    n_side = a.tiles_per_side
    n = n_side * n_side
    W = n_side * T
    tl = np.zeros(n, dtype=gpu.INFLATE_TILE_DTYPE)
    offs, parts, off = [], [], 0
    for s in streams:
        offs.append(off)
        pad = (-len(s)) % 16 + 16
        parts += [s, bytes(pad)]
        off += len(s) + pad
    for k in range(n):
        j = k % len(streams)
        ty, tx = divmod(k, n_side)
        tl[k] = (offs[j], len(streams[j]), T * T, T, 0, 0, T, T, 0, ty * T * W + tx * T)
    comp = np.frombuffer(b"".join(parts), dtype=np.uint8)
    with gpu.Engine(0) as e:
        bufs = [e.upload(comp), e.upload(tl.view(np.uint8)), e.alloc(W * W), e.alloc(4 * n)]
        e0, e1 = e.event_create(), e.event_create()
        if a.diag:
            e.set_option("inflate_diag", a.diag)
        ms = []
        for rep in range(a.reps + 1):
            e.event_record(e0)
            e._chk(gpu.lib().gcn10_gpu_inflate_tiles(e._ctx, bufs[0].ptr, bufs[1].ptr, n, T * T, bufs[2].ptr, W,
                                                     bufs[3].ptr, None), "inflate")
            e.event_record(e1)
            e.event_sync(e1)
            ms.append(e.elapsed_ms(e0, e1))
        status = e.download(bufs[3].ptr, (n,), dtype=np.uint32)
        # spot check
        row = e.download(bufs[2].ptr + 0, (T, W))
        ok = not status.any() and all(np.array_equal(row[:, k * T:(k + 1) * T], tiles[k % len(tiles)]) for k in range(n_side))
        for b in bufs:
            b.close()
    best = min(ms[1:])
    comp_bytes = sum(len(streams[k % len(streams)]) for k in range(n))
    print(json.dumps({"pattern": a.pattern, "level": a.level, "tiles": n, "raw_bytes": n * T * T,
                      "compressed_bytes": comp_bytes, "ratio": round(n * T * T / comp_bytes, 1),
                      "ms": [round(m, 2) for m in ms], "best_ms": round(best, 2),
                      "decoded_GBps": round(n * T * T / best / 1e6, 1), "ok": bool(ok),
                      "host_zlib_ms_per_tile": round(host_inflate_s * 1e3, 2),
                      "host_zlib_GBps_one_core": round(T * T / host_inflate_s / 1e9, 3)}))


if __name__ == "__main__":
    main()

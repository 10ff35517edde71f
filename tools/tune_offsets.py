#!/usr/bin/env python3
"""Does the relative placement of the landcover block and the output raster in HBM matter?
Both streams march through their buffers in lockstep; hipMalloc aligns both to 2 MB, so they hit
the same channel / bank group at the same time.  Times the single-raster strip kernel with the
output shifted by a range of byte offsets.  One JSON line per offset."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from gcn10_amd import gpu, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=36000)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    size = a.size
    eng = gpu.Engine(0)
    eng.set_tables(host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups")))
    esa, gt, coarse, sgt = bench.synth_block(1, size, "iid")
    hs = coarse.shape[0]
    ci, cj = host.build_index_maps(gt, sgt, size, size, hs, hs)
    npix = size * size
    d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
    pad = 64 << 20
    out = eng.alloc(npix + pad)
    e0, e1 = eng.event_create(), eng.event_create()
    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
    eng.sync()
    print(json.dumps({"esa_ptr_mod_2MB": d_esa.ptr % (2 << 20), "out_ptr_mod_2MB": out.ptr % (2 << 20)}))
    offsets = [0, 256, 1024, 4096, 16384, 65536, 131072, 262144, 524288, 1 << 20, (1 << 20) + 4096, 3 << 19,
               (2 << 20) + 65536, 5 << 20, (7 << 20) + 16384, 33 << 20, 0]
    for off in offsets:
        ptrs = [None] * 18
        ptrs[7] = out.ptr + off
        eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 1, 1 << 7, ptrs)
        eng.sync()
        eng.event_record(e0)
        for _ in range(a.reps):
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 1, 1 << 7, ptrs)
        eng.event_record(e1)
        eng.sync()
        ms = eng.elapsed_ms(e0, e1) / a.reps
        print(json.dumps({"out_offset": off, "ms": round(ms, 4), "frac": round(2.0018 * npix / ms / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Times the strip-kernel launch variants on one resident 36000^2 block (GPU only).

    python tools/tune_strip.py [--size 36000] [--reps 10] > gpurun_out/tune.json

Every variant's output is compared (SHA-256 of raster 7 / all selected rasters'
first MB) with the default variant's so a faster-but-wrong variant is flagged.
"""
import argparse
import hashlib
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402
from gcn10_amd import gpu, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=36000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pattern", default="iid")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    size = a.size
    eng = gpu.Engine(0)
    eng.set_tables(host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups")))
    esa, gt, coarse, sgt = bench.synth_block(1, size, a.pattern)
    hs = coarse.shape[0]
    ci, cj = host.build_index_maps(gt, sgt, size, size, hs, hs)
    npix = size * size
    d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
    outs = [eng.alloc(npix) for _ in range(18)]
    ptrs = [o.ptr for o in outs]
    e0, e1 = eng.event_create(), eng.event_create()
    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
    eng.sync()

    def digest(rasters):
        h = hashlib.sha256()
        for r in rasters:
            h.update(eng.download(outs[r].ptr, (1 << 20,)).tobytes())
            h.update(eng.download(outs[r].at(npix - (1 << 20)), (1 << 20,)).tobytes())
        return h.hexdigest()[:16]

    def time_it(cond_mask, table_mask, rasters):
        for r in rasters:
            eng.memset(outs[r].ptr, 0, 1 << 20)
        eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, cond_mask, table_mask, ptrs)
        eng.sync()
        eng.event_record(e0)
        for _ in range(a.reps):
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, cond_mask, table_mask, ptrs)
        eng.event_record(e1)
        eng.sync()
        return eng.elapsed_ms(e0, e1) / a.reps

    results = []
    # prepare_tile timing
    eng.event_record(e0)
    for _ in range(a.reps):
        eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
    eng.event_record(e1)
    eng.sync()
    results.append({"kernel": "expand_x_codes", "ms": eng.elapsed_ms(e0, e1) / a.reps})
    print(json.dumps(results[-1]), flush=True)

    workloads = [("config2", 1, 1 << 7, [7], "ilp1", [1, 2, 4]),
                 ("config4", 3, 0x1FF, list(range(18)), "ilp16", [1, 2]),
                 ("config4-drained", 1, 0x1FF, list(range(9)), "ilp16", [1, 2])]
    ref = {}
    for name, cm, tm, rasters, ilp_key, ilps in workloads:
        alg = gpu.strip_algorithmic_bytes(size, size, hs, hs, cm, tm)
        grids = [4, 8, 16] if not a.quick else [8, 16]
        for ilp, nt, xcd, bpc, pf in itertools.product(ilps, [1, 0], [1, 0], grids, [1, 0]):
            if a.quick and (nt == 0 or xcd == 0):
                continue
            if pf and ilp > 2:
                continue            # pipelined variants exist for 1 and 2 chunks per trip
            eng.set_option("prefetch", pf)
            eng.set_option(ilp_key, ilp)
            eng.set_option("nontemporal", nt)
            eng.set_option("xcd_slabs", xcd)
            eng.set_option("grid_blocks_per_cu", bpc)
            ms = time_it(cm, tm, rasters)
            d = digest(rasters[:3])
            ref.setdefault(name, d)
            rec = {"workload": name, "ilp": ilp, "pf": pf, "nt": nt, "xcd_slabs": xcd, "blocks_per_cu": bpc,
                   "ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 1),
                   "frac": round(alg / ms / 1e6 / 8000, 4), "ok": d == ref[name]}
            results.append(rec)
            print(json.dumps(rec), flush=True)
    best = {}
    for r in results:
        if "workload" in r and r["ok"]:
            if r["workload"] not in best or r["ms"] < best[r["workload"]]["ms"]:
                best[r["workload"]] = r
    print(json.dumps({"best": best}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How much does the choice of output allocation matter for the strip kernels?  (GPU only.)

Allocates N candidate rasters (separate hipMallocs of one 36000^2 block each), then times, in
interleaved rounds with events carried by the dispatch:
  * the plain copy landcover -> candidate i,
  * the config-2 kernel (one raster) writing candidate i,
  * the config-4 kernel (18 rasters) writing K random 18-subsets of the candidates.
Prints one JSON line per case plus a summary (min / median / max).
"""
import argparse
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
    ap.add_argument("--candidates", type=int, default=36)
    ap.add_argument("--subsets", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--per-round", type=int, default=4)
    a = ap.parse_args()
    size = a.size
    eng = gpu.Engine(0)
    eng.set_tables(host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups")))
    esa, gt, coarse, sgt = bench.synth_block(1, size, "iid")
    hs = coarse.shape[0]
    ci, cj = host.build_index_maps(gt, sgt, size, size, hs, hs)
    npix = size * size
    d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
    cands = [eng.alloc(npix) for _ in range(a.candidates)]
    ev = [(eng.event_create(), eng.event_create()) for _ in range(a.per_round)]
    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
    eng.sync()
    nb = npix - npix % 16
    rng = np.random.default_rng(7)
    cases = []
    for i, c in enumerate(cands):
        cases.append({"kind": "copy", "i": i, "dist_MiB": round((c.ptr - d_esa.ptr) / 2**20, 2)})
        cases.append({"kind": "config2", "i": i, "dist_MiB": round((c.ptr - d_esa.ptr) / 2**20, 2)})
    for s in range(a.subsets):
        pick = sorted(rng.choice(a.candidates, size=18, replace=False).tolist()) if s else list(range(18))
        cases.append({"kind": "config4", "subset": pick})

    def launch(c, timed=None):
        if timed is not None:
            eng.time_next_strip(*timed)
        if c["kind"] == "copy":
            eng.stream_copy(d_esa.ptr, cands[c["i"]].ptr, nb)
        elif c["kind"] == "config2":
            ptrs = [None] * 18
            ptrs[7] = cands[c["i"]].ptr
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 1, 1 << 7, ptrs)
        else:
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, [cands[j].ptr for j in c["subset"]])

    for c in cases:
        c["ms"] = []
    for _ in range(a.rounds):
        for c in cases:
            launch(c)
            for i in range(a.per_round):
                launch(c, ev[i])
            eng.sync()
            c["ms"] += [eng.elapsed_ms(*ev[i]) for i in range(a.per_round)]
    summ = {}
    for c in cases:
        ms = np.array(c.pop("ms"))
        c["median_ms"] = round(float(np.median(ms)), 4)
        c["min_ms"] = round(float(ms.min()), 4)
        print(json.dumps(c), flush=True)
        summ.setdefault(c["kind"], []).append(c["median_ms"])
    print(json.dumps({"summary": {k: {"n": len(v), "min": min(v), "median": float(np.median(v)), "max": max(v)}
                                  for k, v in summ.items()}, "esa_ptr": hex(d_esa.ptr)}))


if __name__ == "__main__":
    main()

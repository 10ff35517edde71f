#!/usr/bin/env python3
"""The single-raster strip kernel on BASELINE config 2 with and without gcn10_gpu_prepare_tile in front of every
launch (is the strip kernel slower right after the soil codes were rewritten?), compact soil words on and off, interleaved
rounds.  (GPU only.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from gcn10_amd import gpu, host  # noqa: E402

size = int(os.environ.get("SIZE", "36000"))
npix = size * size
eng = gpu.Engine(0)
eng.set_tables(host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups")))
esa, gt, coarse, soil_gt = bench.synth_block(1, size, "iid")
hs = coarse.shape[0]
ci, cj = host.build_index_maps(gt, soil_gt, size, size, hs, hs)
d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
outs = {"plain": eng.alloc(npix), "plain 2": eng.alloc(npix)}
eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
ev = [(eng.event_create(), eng.event_create()) for _ in range(5)]
variants = [("flat ilp2 pf bpc8", dict(ilp1=2, prefetch=1, grid_blocks_per_cu=8)),
            ("flat ilp2 pf bpc8, soil BYTES", dict(ilp1=2, prefetch=1, grid_blocks_per_cu=8, compact_soil=0)),
            ("flat ilp4 bpc16, soil BYTES", dict(ilp1=4, prefetch=0, grid_blocks_per_cu=16, compact_soil=0)),
            ("flat ilp2 pf bpc16", dict(ilp1=2, prefetch=1, grid_blocks_per_cu=16)),
            ("flat ilp2 pf bpc16, soil BYTES", dict(ilp1=2, prefetch=1, grid_blocks_per_cu=16, compact_soil=0)),
            ("flat ilp1 pf bpc16", dict(ilp1=1, prefetch=1, grid_blocks_per_cu=16)),
            ("flat ilp4 bpc8", dict(ilp1=4, prefetch=0, grid_blocks_per_cu=8)),
            ("flat ilp2 pf bpc8, prepare_tile before every launch", dict(ilp1=2, prefetch=1, grid_blocks_per_cu=8, PREP=1)),
            ("flat ilp4 bpc16", dict(ilp1=4, prefetch=0, grid_blocks_per_cu=16)),
            ("flat ilp4 bpc16, prepare_tile before every launch", dict(ilp1=4, prefetch=0, grid_blocks_per_cu=16, PREP=1)),
            ("copy", None)]
_unused = [("flat ilp2 pf bpc8", dict(tile_rows=0, ilp1=2, prefetch=1, grid_blocks_per_cu=8)),
            ("flat ilp4 bpc16", dict(tile_rows=0, ilp1=4, prefetch=0, grid_blocks_per_cu=16)),
            ("tile8 bpc8", dict(tile_rows=8, grid_blocks_per_cu=8)),
            ("tile8 bpc12", dict(tile_rows=8, grid_blocks_per_cu=12)),
            ("tile8 bpc16", dict(tile_rows=8, grid_blocks_per_cu=16)),
            ("tile4 bpc8", dict(tile_rows=4, grid_blocks_per_cu=8)),
            ("tile4 bpc16", dict(tile_rows=4, grid_blocks_per_cu=16)),
            ("tile8 bpc16 no slabs", dict(tile_rows=8, grid_blocks_per_cu=16, xcd_slabs=0)),
            ("tile4 bpc32", dict(tile_rows=4, grid_blocks_per_cu=32)),
            ("tile8 bpc16 NO SOIL LOADS", dict(tile_rows=8, grid_blocks_per_cu=16, fused_diag=1)),
            ("tile4 bpc16 NO SOIL LOADS", dict(tile_rows=4, grid_blocks_per_cu=16, fused_diag=1)),
            ("copy", None)]
res = {}
for rnd in range(4):
    for where, buf in outs.items():
        ptrs = [None] * 18
        ptrs[K := 1] = buf.ptr
        for name, opts in variants:
            eng.set_option("defaults", 0)
            if opts:
                for k, v in opts.items():
                    if k != "PREP":
                        eng.set_option(k, v)

            def go():
                if opts is None:
                    eng.stream_copy(d_esa.ptr, buf.ptr, npix - npix % 16)
                else:
                    if opts.get("PREP"):
                        eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
                    eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 1, 1 << K, ptrs)
            go()
            for k in range(5):
                eng.time_next_strip(*ev[k])
                go()
            eng.sync()
            ms = sorted(eng.elapsed_ms(*ev[k]) for k in range(5))[2]
            res.setdefault((where, name), []).append(ms)
            if rnd == 0 and opts:
                res.setdefault(("kernel", name), eng.last_kernel_name())
for (where, name), v in res.items():
    if where == "kernel":
        continue
    print(json.dumps({"raster": where, "variant": name, "median_ms": [round(x, 4) for x in v], "best": round(min(v), 4),
                      "kernel": res.get(("kernel", name), "stream_copy_kernel")}))

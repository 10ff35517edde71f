#!/usr/bin/env python3
"""Times gcn10_gpu_prepare_tile (expand_x_codes) on a 36000-wide block: 50 calls between two events."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gcn10_amd import gpu, host
size = int(sys.argv[1]) if len(sys.argv) > 1 else 36000
eng = gpu.Engine(0)
esa, gt, coarse, sgt = bench.synth_block(1, 2048, "iid")
hs = size // 25
import numpy as np
coarse = np.random.default_rng(1).choice(bench.HSG_CODES, size=(hs, hs)).astype(np.uint8)
gt = [0.0, 3.0 / size, 0.0, 3.0, 0.0, -3.0 / size]
sgt = [0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs]
ci, cj = host.build_index_maps(gt, sgt, size, size, hs, hs)
d_coarse, d_ci = eng.upload(coarse), eng.upload(ci)
e0, e1 = eng.event_create(), eng.event_create()
for _ in range(5):
    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
eng.sync()
eng.event_record(e0)
for _ in range(50):
    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
eng.event_record(e1)
eng.sync()
print(json.dumps({"prepare_tile_us": round(eng.elapsed_ms(e0, e1) / 50 * 1e3, 2), "W": size, "soil": [hs, hs]}))

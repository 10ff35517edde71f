"""Corrupted inputs must be rejected, never crash: a small deterministic mutation run over the
GeoTIFF and shapefile readers (the same loop runs clean for thousands of mutations under
AddressSanitizer + UBSan: `make asan-host`)."""
import random

import numpy as np

from gcn10_amd import host
from tests import tiffutil


def _mutate(data, rnd):
    d = bytearray(data)
    if rnd.random() < 0.3:
        return d[:rnd.randrange(1, len(d))]
    for _ in range(rnd.randrange(1, 6)):
        d[rnd.randrange(len(d))] = rnd.randrange(256)
    return d


def test_mutated_tiffs_do_not_crash(tmp_path):
    rnd = random.Random(1)
    img = np.random.default_rng(0).integers(0, 5, size=(70, 90), dtype=np.uint8) * 40
    variants = [dict(compression=5, tile=(32, 32)), dict(compression=8, rows_per_strip=8),
                dict(compression=32773), dict(compression=5, predictor=2),
                dict(compression=8, bigtiff=True, tile=(32, 32))]
    opened = rejected = 0
    for vi, kw in enumerate(variants):
        p = tmp_path / ("v%d.tif" % vi)
        tiffutil.write_tiff(str(p), img, gt=[0, 1, 0, 0, 0, -1], **kw)
        data = p.read_bytes()
        for _ in range(60):
            q = tmp_path / "m.tif"
            q.write_bytes(bytes(_mutate(data, rnd)))
            try:
                with host.Raster(str(q)) as r:
                    opened += 1
                    if 0 < r.xsize * r.ysize < 10_000_000:
                        r.read(0, 0, min(r.xsize, 90), min(r.ysize, 70))
            except host.HostError:
                rejected += 1
    assert opened > 0 and rejected > 0


def test_mutated_shapefiles_do_not_crash(tmp_path):
    rnd = random.Random(2)
    base = str(tmp_path / "b")
    tiffutil.write_block_shapefile(base, [(i, float(i), 0.0, float(i) + 3, 3.0) for i in range(1, 40)])
    shp, dbf = open(base + ".shp", "rb").read(), open(base + ".dbf", "rb").read()
    for it in range(200):
        s, d = (bytes(_mutate(shp, rnd)), dbf) if it % 2 else (shp, bytes(_mutate(dbf, rnd)))
        open(str(tmp_path / "m.shp"), "wb").write(s)
        open(str(tmp_path / "m.dbf"), "wb").write(d)
        try:
            ids, bbox = host.read_blocks_shapefile(str(tmp_path / "m.shp"))
            assert len(ids) == len(bbox)
        except host.HostError:
            pass

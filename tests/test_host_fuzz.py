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


def test_read_plans_of_mutated_tiffs_stay_inside_the_file(tmp_path):
    """The plan handed to the GPU decoder names file ranges: for a damaged directory they must be
    refused or lie inside the file (Raster.plan reads them with pread)."""
    rnd = random.Random(3)
    img = np.random.default_rng(1).integers(0, 5, size=(70, 90), dtype=np.uint8) * 40
    planned = refused = 0
    for vi, kw in enumerate([dict(tile=(32, 32)), dict(rows_per_strip=8), dict(tile=(16, 16), bigtiff=True)]):
        p = tmp_path / ("p%d.tif" % vi)
        tiffutil.write_tiff(str(p), img, gt=[0, 1, 0, 0, 0, -1], compression=8, **kw)
        data = p.read_bytes()
        for _ in range(80):
            q = tmp_path / "m.tif"
            q.write_bytes(bytes(_mutate(data, rnd)))
            size = q.stat().st_size
            try:
                with host.Raster(str(q)) as r:
                    if not (0 < r.xsize * r.ysize < 10_000_000):
                        continue
                    plan = r.plan(0, 0, min(r.xsize, 90), min(r.ysize, 70))
                    if plan is None:
                        continue
                    planned += 1
                    for c in plan[0]:
                        assert len(c["data"]) <= size
                        assert c["src_x"] + c["copy_w"] <= c["chunk_w"] and c["src_y"] + c["copy_h"] <= c["rows"]
            except host.HostError:
                refused += 1
    assert planned > 0 and refused > 0


def test_mutated_vrts_do_not_crash(tmp_path):
    rnd = random.Random(4)
    img = np.random.default_rng(2).integers(1, 6, size=(40, 60), dtype=np.uint8) * 10
    tiffutil.write_tiff(str(tmp_path / "a.tif"), img[:, :30], compression=8, tile=(16, 16))
    tiffutil.write_tiff(str(tmp_path / "b.tif"), img[:, 30:], compression=5, rows_per_strip=7)
    src = ""
    for name, dx in (("a.tif", 0), ("b.tif", 30)):
        src += ('<ComplexSource resampling="nearest"><SourceFilename relativeToVRT="1">%s</SourceFilename>'
                '<SourceBand>1</SourceBand><SrcRect xOff="0" yOff="0" xSize="30" ySize="40" />'
                '<DstRect xOff="%d" yOff="0" xSize="30" ySize="40" /><NODATA>0</NODATA></ComplexSource>\\n' % (name, dx))
    good = ('<VRTDataset rasterXSize="60" rasterYSize="40">\\n<GeoTransform> 0.0, 1.0, 0.0, 0.0, 0.0, -1.0</GeoTransform>\\n'
            '<VRTRasterBand dataType="Byte" band="1"><NoDataValue>0</NoDataValue>\\n%s</VRTRasterBand></VRTDataset>\\n' % src)
    (tmp_path / "g.vrt").write_text(good)
    with host.Raster(str(tmp_path / "g.vrt")) as r:
        assert np.array_equal(r.read(0, 0, 60, 40), img)
    opened = 0
    for _ in range(300):
        (tmp_path / "m.vrt").write_bytes(bytes(_mutate(good.encode(), rnd)))
        try:
            with host.Raster(str(tmp_path / "m.vrt")) as r:
                opened += 1
                if 0 < r.xsize * r.ysize < 1_000_000:
                    try:
                        r.read(0, 0, min(r.xsize, 60), min(r.ysize, 40))
                        r.plan(0, 0, min(r.xsize, 60), min(r.ysize, 40))
                    except host.HostError:
                        pass
        except host.HostError:
            pass
    assert opened > 0

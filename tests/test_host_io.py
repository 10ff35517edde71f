"""Host I/O of the gcn10 program: config, logs, block index, GeoTIFF/VRT read, GeoTIFF write.
Independent checkers: Pillow (libtiff) for TIFF bytes, a Python TIFF/LZW producer for inputs."""
import os
import re

import numpy as np
import pytest
from PIL import Image

from gcn10_amd import host
from oracle import cn_oracle_c as oc
from tests import tiffutil
from tests.conftest import GOLDEN

Image.MAX_IMAGE_PIXELS = None


# ---- config (src/config.c) ------------------------------------------------

def test_config_reference_file_shape(tmp_path):
    p = tmp_path / "config.txt"
    p.write_text("# comment\n\nhysogs_data_path=../../hsg/HYSOGs250m_4326_lzw.tif\n"
                 "  esa_data_path =  ../../landcover/esa_worldcover_2021.vrt  \n"
                 "blocks_shp_path=../../blocks/esa_extent_blocks.shp\nlookup_table_path=../../lookups\n"
                 "log_dir=logs/\nunknown_key=1\nno equals sign here\nlog_dir=logs2/\n")
    c = host.parse_config(str(p))
    assert c["hysogs_data_path"] == "../../hsg/HYSOGs250m_4326_lzw.tif"
    assert c["esa_data_path"] == "../../landcover/esa_worldcover_2021.vrt"     # trimmed
    assert c["log_dir"] == "logs2/"                                            # last one wins
    assert c["gpus"] == 0 and c["esa_tile_dir"] is None and c["gpu_deflate"] == 2 and c["gpu_inflate"] == 1


def test_config_optional_keys_and_errors(tmp_path):
    p = tmp_path / "c.txt"
    base = "hysogs_data_path=a\nesa_data_path=b\nblocks_shp_path=c\nlookup_table_path=d\nlog_dir=e\n"
    p.write_text(base + "workers_per_gpu=3\ngpus=4\nstrip_rows=512\nio_threads=3\ndeflate_level=1\nesa_tile_dir=/x\ngpu_deflate=0\ngpu_inflate=0\n")
    c = host.parse_config(str(p))
    assert c["gpu_deflate"] == 0 and c["workers_per_gpu"] == 3 and c["gpu_inflate"] == 0
    assert (c["gpus"], c["strip_rows"], c["io_threads"], c["deflate_level"], c["esa_tile_dir"]) == \
        (4, 512, 3, 1, "/x")
    p.write_text("hysogs_data_path=a\nesa_data_path=b\n")
    with pytest.raises(host.HostError, match="missing one of: hysogs_data_path"):      # src/config.c:109
        host.parse_config(str(p))
    with pytest.raises(host.HostError, match="cannot open config"):                    # src/config.c:52
        host.parse_config(str(tmp_path / "nope.txt"))


# ---- logging (src/log.c) ----------------------------------------------------

def test_log_format(tmp_path):
    d = tmp_path / "logs"
    lg = host.Log(str(d), 3)
    lg.message("INFO", "processing block 2234")
    lg.message("ERROR", "block 7 not found", also_console=True)
    lg.message(None, None)
    lg.close()
    lines = (d / "rank_3.log").read_text().splitlines()
    ts = r"\[\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\]"
    assert re.fullmatch(ts + r" \[rank 3\] logging started", lines[0])                 # src/log.c:112
    assert re.fullmatch(ts + r" \[INFO\] \[rank 3\] processing block 2234", lines[1])  # src/log.c:159
    assert re.fullmatch(ts + r" \[ERROR\] \[rank 3\] block 7 not found", lines[2])
    assert re.fullmatch(ts + r" \[INFO\] \[rank 3\] ", lines[3])
    assert re.fullmatch(ts + r" \[rank 3\] logging finished", lines[4])                # src/log.c:264
    lg = host.Log(str(d), 3)        # append mode (src/log.c:76)
    lg.close()
    assert len((d / "rank_3.log").read_text().splitlines()) == 7


# ---- block index ------------------------------------------------------------

def test_block_list_parsing(tmp_path):
    p = tmp_path / "blocks.txt"
    p.write_text("2234\n2261 2256\t2257\n\n 2290 x 99\n")
    assert host.read_block_list(str(p)) == [2234, 2261, 2256, 2257, 2290]    # stops at 'x' (src/raster.c:46)
    p.write_text("")
    assert host.read_block_list(str(p)) == []
    assert host.read_block_list(str(tmp_path / "missing.txt")) is None
    p.write_text(" ".join(str(i) for i in range(1000)))
    assert host.read_block_list(str(p)) == list(range(1000))


def test_reference_block_shapefile():
    ids, bbox = host.read_blocks_shapefile(os.path.join(GOLDEN, "blocks", "esa_extent_blocks.shp"))
    assert len(ids) == 2651 and min(ids) == 3 and max(ids) == 2653 and len(set(ids)) == 2651
    assert np.all(bbox[:, 2] - bbox[:, 0] == 3.0) and np.all(bbox[:, 3] - bbox[:, 1] == 3.0)
    assert np.all(bbox == np.round(bbox))                      # integer degrees
    i = ids.index(2234)                                        # first id of src/test/blocks.txt
    assert bbox[i].tolist() == [-114.0, 39.0, -111.0, 42.0]
    # every block gives the 36001 / 36000 windows of SURVEY.md section 7 on the real VRT grid
    vrt_gt = [-180.0, 8.3333333333330430e-05, 0.0, 84.0, 0.0, -8.3333333333330430e-05]
    sizes = set()
    for b in bbox[::50]:
        w = host.raster_window(vrt_gt, 4320000, 1728000, b.tolist())
        assert w == oc.window(vrt_gt, 4320000, 1728000, b.tolist())
        sizes.add((w[2], w[3]))
    assert sizes <= {(36001, 36001), (36000, 36001), (36001, 36000), (36000, 36000)}


@pytest.mark.parametrize("shape_type", [5, 15])
def test_synthetic_block_shapefile(tmp_path, shape_type):
    blocks = [(7, -3.0, 0.0, 0.0, 3.0), (12, 0.0, 0.0, 3.0, 3.0), (1000, 10.5, -20.25, 11.0, -19.0), (7, 1, 1, 2, 2)]
    base = str(tmp_path / "b")
    tiffutil.write_block_shapefile(base, blocks, shape_type)
    ids, bbox = host.read_blocks_shapefile(base + ".shp")
    assert ids == [7, 12, 1000, 7]
    assert np.array_equal(bbox, np.array([b[1:] for b in blocks], dtype=np.float64))
    with pytest.raises(host.HostError, match="ogr open failed"):           # src/cn.c:157
        host.read_blocks_shapefile(str(tmp_path / "none.shp"))


# ---- GeoTIFF reader ---------------------------------------------------------

def _img(seed, H, W):
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 12, size=((H + 7) // 8, (W + 7) // 8), dtype=np.uint8) * 10
    img = np.repeat(np.repeat(small, 8, axis=0), 8, axis=1)[:H, :W].copy()
    noise = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    return np.where(noise < 40, noise, img).astype(np.uint8)


GT = [-111.0, 0.01, 0.0, 39.0, 0.0, -0.02]

VARIANTS = [
    dict(compression=1), dict(compression=1, rows_per_strip=7), dict(compression=8, rows_per_strip=16),
    dict(compression=32946), dict(compression=5), dict(compression=5, rows_per_strip=5),
    dict(compression=5, predictor=2), dict(compression=8, predictor=2, rows_per_strip=3),
    dict(compression=32773, rows_per_strip=9), dict(compression=1, tile=(16, 16)),
    dict(compression=8, tile=(32, 16)), dict(compression=5, tile=(64, 64)),
    dict(compression=5, tile=(16, 32), predictor=2), dict(compression=8, big_endian=True),
    dict(compression=5, bigtiff=True, tile=(32, 32)), dict(compression=1, bigtiff=True, big_endian=True),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=[str(sorted(v.items())) for v in VARIANTS])
def test_tiff_reader_variants(tmp_path, kw):
    img = _img(len(str(kw)), 75, 101)
    p = str(tmp_path / "t.tif")
    tiffutil.write_tiff(p, img, gt=GT, **kw)
    if not kw.get("bigtiff"):
        # the producer itself is checked by libtiff (through Pillow)
        assert np.array_equal(np.array(Image.open(p)), img)
    with host.Raster(p) as r:
        assert (r.xsize, r.ysize) == (101, 75)
        assert r.gt == GT
        assert np.array_equal(r.read(0, 0, 101, 75), img)
        assert np.array_equal(r.read(13, 9, 50, 41), img[9:50, 13:63])
        assert np.array_equal(r.read(100, 74, 1, 1), img[74:, 100:])
        with pytest.raises(host.HostError):
            r.read(90, 0, 20, 5)


@pytest.mark.parametrize("kw", [dict(compression=5, rows_per_strip=8), dict(compression=8, rows_per_strip=3, predictor=2),
                                dict(compression=8, tile=(256, 64)), dict(compression=1, rows_per_strip=8)],
                         ids=["lzw strips", "deflate strips + predictor", "deflate tiles", "raw strips (row reads, no cache)"])
def test_windows_that_use_a_small_part_of_their_chunks_go_through_the_chunk_cache(tmp_path, kw):
    """Round 3: a window that uses at most a quarter of a compressed chunk (the soil raster: full-width strips, a
    block needs 1/120 of each) reads it through a per-process cache of decoded chunks, shared by every handle of the
    file.  Narrow windows side by side, from two handles, a second pass (all hits), windows that straddle chunk
    borders and the ragged last strip: every read equals the source, and the cache reports the hits."""
    import ctypes as C
    img = _img(77, 61, 1500)
    p = str(tmp_path / "wide.tif")
    tiffutil.write_tiff(p, img, gt=GT, **kw)
    L = host.lib()
    L.gcn10_tiff_cache_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    L.gcn10_tiff_cache_stats.restype = None
    h0, m0, b0 = C.c_uint64(), C.c_uint64(), C.c_size_t()
    L.gcn10_tiff_cache_stats(C.byref(h0), C.byref(m0), C.byref(b0))
    with host.Raster(p) as a, host.Raster(p) as b:
        for rep in range(2):
            for k, x in enumerate(range(0, 1500 - 40, 97)):
                r = a if k % 2 else b
                assert np.array_equal(r.read(x, 5, 40, 50), img[5:55, x:x + 40])
        assert np.array_equal(a.read(1460, 58, 40, 3), img[58:61, 1460:1500])      # the ragged last strip
        assert np.array_equal(a.read(0, 0, 1500, 61), img)                           # a window that uses all of them: not cached
    h1, m1, b1 = C.c_uint64(), C.c_uint64(), C.c_size_t()
    L.gcn10_tiff_cache_stats(C.byref(h1), C.byref(m1), C.byref(b1))
    if kw["compression"] != 1:
        assert m1.value > m0.value and h1.value - h0.value > 4 * (m1.value - m0.value)
        assert b1.value > 0


def test_tiff_reader_pillow_written_files(tmp_path):
    img = _img(5, 300, 257)
    for comp in ("raw", "tiff_lzw", "tiff_adobe_deflate", "packbits"):
        p = str(tmp_path / ("p_%s.tif" % comp))
        Image.fromarray(img).save(p, compression=None if comp == "raw" else comp)
        with host.Raster(p) as r:
            assert np.array_equal(r.read(0, 0, 257, 300), img), comp


def test_tiff_reader_pixel_is_point_and_errors(tmp_path):
    img = _img(1, 10, 10)
    p = str(tmp_path / "pt.tif")
    tiffutil.write_tiff(p, img, gt=GT, pixel_is_point=True)
    with host.Raster(p) as r:       # GDAL shifts the origin by half a pixel for PixelIsPoint
        assert r.gt[0] == GT[0] - 0.5 * GT[1] and r.gt[3] == GT[3] - 0.5 * GT[5]
    (tmp_path / "junk.tif").write_bytes(b"not a tiff at all")
    with pytest.raises(host.HostError, match="gdal open failed"):          # src/raster.c:121
        host.Raster(str(tmp_path / "junk.tif"))
    with pytest.raises(host.HostError, match="gdal open failed"):
        host.Raster(str(tmp_path / "missing.tif"))
    Image.fromarray(img.astype(np.uint16) * 100).save(str(tmp_path / "u16.tif"))
    with pytest.raises(host.HostError, match="only Byte rasters"):
        host.Raster(str(tmp_path / "u16.tif"))


def test_lzw_long_runs_and_table_resets(tmp_path):
    # > 4094 dictionary entries forces ClearCodes; long runs hit the KwKwK case
    rng = np.random.default_rng(3)
    img = np.concatenate([np.zeros((40, 512), np.uint8), rng.integers(0, 256, (200, 512), dtype=np.uint8),
                          np.full((40, 512), 200, np.uint8)])
    p = str(tmp_path / "lzw.tif")
    tiffutil.write_tiff(p, img, compression=5)
    assert np.array_equal(np.array(Image.open(p)), img)
    with host.Raster(p) as r:
        assert np.array_equal(r.read(0, 0, 512, 280), img)


# ---- VRT ------------------------------------------------------------------------

def _make_vrt(tmp_path, vsicurl=False):
    a, b = _img(11, 64, 64), _img(12, 64, 64)
    a[a == 0] = 10
    b[:8, :8] = 0                       # NODATA region of the second source
    tiffutil.write_tiff(str(tmp_path / "tile_a.tif"), a, compression=8, tile=(32, 32))
    tiffutil.write_tiff(str(tmp_path / "tile_b.tif"), b, compression=5)
    pre = "/vsicurl/https://example.invalid/v200/" if vsicurl else ""
    rel = "0" if vsicurl else "1"
    vrt = """<VRTDataset rasterXSize="160" rasterYSize="96">
  <GeoTransform> -1.8000000000000000e+02,  8.3333333333330430e-05,  0.0000000000000000e+00,  8.4000000000000000e+01,  0.0000000000000000e+00, -8.3333333333330430e-05</GeoTransform>
  <VRTRasterBand dataType="Byte" band="1">
    <NoDataValue>0</NoDataValue>
    <ComplexSource resampling="nearest">
      <SourceFilename relativeToVRT="%s">%stile_a.tif</SourceFilename>
      <SourceBand>1</SourceBand>
      <SrcRect xOff="0" yOff="0" xSize="64" ySize="64" />
      <DstRect xOff="16" yOff="8" xSize="64" ySize="64" />
      <NODATA>0</NODATA>
    </ComplexSource>
    <ComplexSource resampling="nearest">
      <SourceFilename relativeToVRT="%s">%stile_b.tif</SourceFilename>
      <SourceBand>1</SourceBand>
      <SrcRect xOff="0" yOff="0" xSize="64" ySize="64" />
      <DstRect xOff="72" yOff="20" xSize="64" ySize="64" />
      <NODATA>0</NODATA>
    </ComplexSource>
  </VRTRasterBand>
</VRTDataset>
""" % (rel, pre, rel, pre)
    (tmp_path / "m.vrt").write_text(vrt)
    exp = np.zeros((96, 160), np.uint8)
    exp[8:72, 16:80] = a
    sub = exp[20:84, 72:136]
    exp[20:84, 72:136] = np.where(b != 0, b, sub)
    return exp


def test_vrt_mosaic(tmp_path):
    exp = _make_vrt(tmp_path)
    with host.Raster(str(tmp_path / "m.vrt")) as r:
        assert (r.xsize, r.ysize) == (160, 96)
        assert r.gt == [-180.0, 8.3333333333330430e-05, 0.0, 84.0, 0.0, -8.3333333333330430e-05]
        assert np.array_equal(r.read(0, 0, 160, 96), exp)
        assert np.array_equal(r.read(60, 10, 50, 40), exp[10:50, 60:110])
        assert not r.read(140, 0, 20, 8).any()          # nothing there: NoData 0


def test_vrt_vsicurl_sources_need_a_local_mirror(tmp_path):
    exp = _make_vrt(tmp_path, vsicurl=True)
    with host.Raster(str(tmp_path / "m.vrt"), tile_dir=str(tmp_path)) as r:
        assert np.array_equal(r.read(0, 0, 160, 96), exp)
    with host.Raster(str(tmp_path / "m.vrt")) as r:     # no mirror: the read fails, loudly
        with pytest.raises(host.HostError, match="gdal open failed"):
            r.read(0, 0, 160, 96)
        assert not r.read(140, 0, 20, 8).any()          # windows that touch no source still work


def test_reference_vrt_header_parses():
    ref = "/root/reference/landcover/esa_worldcover_2021.vrt"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    with host.Raster(ref) as r:
        assert (r.xsize, r.ysize) == (4320000, 1728000)
        assert r.gt == [-180.0, 8.3333333333330430e-05, 0.0, 84.0, 0.0, -8.3333333333330430e-05]


# ---- GeoTIFF writer (src/raster.c:192-227) -----------------------------------------

@pytest.mark.parametrize("shape", [(1, 1), (256, 256), (257, 255), (300, 700), (513, 1025)])
def test_save_raster_roundtrip(tmp_path, shape):
    H, W = shape
    img = _img(H + W, H, W)
    p = str(tmp_path / "out.tif")
    host.save_raster(img, GT, p)
    im = Image.open(p)
    assert np.array_equal(np.array(im), img)                        # libtiff decodes our file
    tags = im.tag_v2
    assert tags[259] == 8 and tags[322] == 256 and tags[323] == 256     # DEFLATE, 256x256 tiles
    assert tags[258] == (8,) and tags[277] == 1 and tags[339] in (1, (1,))
    assert tuple(tags[33550]) == (GT[1], -GT[5], 0.0)
    assert tuple(tags[33922]) == (0.0, 0.0, 0.0, GT[0], GT[3], 0.0)
    assert 4326 in tags[34735]
    assert 42113 not in tags                                        # no NoData tag, like the reference
    with host.Raster(p) as r:
        assert r.gt == GT and (r.xsize, r.ysize) == (W, H)
        assert np.array_equal(r.read(0, 0, W, H), img)


def test_save_raster_copies_input_georef(tmp_path):
    img = _img(2, 40, 40)
    keys = [1, 1, 0, 4, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326, 2054, 0, 1, 9102]
    tiffutil.write_tiff(str(tmp_path / "in.tif"), img, gt=GT, geokeys=keys)
    with host.Raster(str(tmp_path / "in.tif")) as r:
        host.save_raster(img, r.gt, str(tmp_path / "out.tif"), georef_ptr=r.georef_ptr())
    assert tuple(Image.open(str(tmp_path / "out.tif")).tag_v2[34735]) == tuple(keys)


def test_save_raster_errors(tmp_path):
    with pytest.raises(host.HostError, match="write error"):
        host.save_raster(_img(1, 4, 4), GT, str(tmp_path / "no_such_dir" / "x.tif"))


# ---- read plans for the GPU decoder (csrc/host/raster.c, tiff.c) ----------------------------

def _assemble(plan, W, H):
    """What gcn10_gpu_inflate_tiles is asked to do with a plan, done here with stock zlib and numpy: DEFLATE
    chunks inflate to chunk_w x rows pixels; raw chunks ARE out_len bytes of pixels, staged from the first
    wanted row (and, without a predictor, the first wanted pixel) on; predictor 2 sums every chunk row."""
    import zlib
    chunks, covered, max_bytes = plan
    out = np.zeros((H, W), np.uint8)
    seen = 0
    for c in chunks:
        cw = c["chunk_w"]
        if c["flags"] & 1:                                  # GCN10_TILE_RAW
            assert len(c["data"]) == c["out_len"]
            raw = np.frombuffer(c["data"], np.uint8)
            n_rows = -(-raw.size // cw)
            t = np.zeros(n_rows * cw, np.uint8)
            t[:raw.size] = raw
        else:
            raw = np.frombuffer(zlib.decompress(c["data"]), np.uint8)
            assert raw.size >= cw * c["rows"] and cw * c["rows"] <= max_bytes and c["out_len"] == cw * c["rows"]
            t = raw[:cw * c["rows"]].copy()
        t = t.reshape(-1, cw)
        if c["flags"] & 2:                                  # GCN10_TILE_PREDICTOR2
            t = np.cumsum(t, axis=1, dtype=np.uint64).astype(np.uint8)
        # the last wanted pixel lies inside what was staged
        assert (c["src_y"] + c["copy_h"] - 1) * cw + c["src_x"] + c["copy_w"] <= c["out_len"]
        out[c["dst_y"]:c["dst_y"] + c["copy_h"], c["dst_x"]:c["dst_x"] + c["copy_w"]] = \
            t[c["src_y"]:c["src_y"] + c["copy_h"], c["src_x"]:c["src_x"] + c["copy_w"]]
        seen += c["copy_w"] * c["copy_h"]
    assert seen == covered
    return out


@pytest.mark.parametrize("kw", [dict(tile=(32, 16)), dict(tile=(64, 64)), dict(rows_per_strip=7), dict(),
                                dict(tile=(16, 16), bigtiff=True), dict(rows_per_strip=3, big_endian=True)],
                         ids=str)
def test_read_plan_of_deflate_files_reassembles_every_window(tmp_path, kw):
    """What the plan says (chunk bytes, clipping, placement), decoded here with stock zlib, equals
    what the host reader returns for the same window."""
    img = _img(7, 75, 101)
    p = str(tmp_path / "t.tif")
    tiffutil.write_tiff(p, img, gt=GT, compression=8, **kw)
    with host.Raster(p) as r:
        for (x, y, w, h) in [(0, 0, 101, 75), (13, 9, 50, 41), (100, 74, 1, 1), (31, 15, 2, 2), (0, 70, 101, 5)]:
            plan = r.plan(x, y, w, h)
            assert plan is not None and plan[1] == w * h
            assert np.array_equal(_assemble(plan, w, h), img[y:y + h, x:x + w]), (x, y, w, h)
        with pytest.raises(host.HostError):
            r.plan(90, 0, 20, 5)


@pytest.mark.parametrize("kw", [dict(compression=5), dict(compression=5, predictor=2),
                                dict(compression=32773, rows_per_strip=9)], ids=str)
def test_read_plan_declines_what_the_gpu_decoder_does_not_take(tmp_path, kw):
    p = str(tmp_path / "t.tif")
    tiffutil.write_tiff(p, _img(8, 40, 60), gt=GT, **kw)
    with host.Raster(p) as r:
        assert r.plan(0, 0, 60, 40) is None


@pytest.mark.parametrize("kw", [dict(compression=1, tile=(32, 16)), dict(compression=1, tile=(64, 64)),
                                dict(compression=1, rows_per_strip=7), dict(compression=1),
                                dict(compression=8, predictor=2, tile=(32, 32)),
                                dict(compression=8, predictor=2, rows_per_strip=11),
                                dict(compression=8, predictor=2, tile=(16, 16), bigtiff=True)], ids=str)
def test_read_plan_of_raw_and_predictor2_files_reassembles_every_window(tmp_path, kw):
    """Round 3: uncompressed chunks are planned like DEFLATE ones (staged from the first wanted row / pixel
    on, untiled on the GPU) and TIFF predictor 2 is a flag of the chunk (summed back on the GPU), for raw
    and DEFLATE chunks alike.  The plan, carried out here as the GPU side is asked to, equals the host reader."""
    img = _img(17, 75, 101)
    p = str(tmp_path / "t.tif")
    tiffutil.write_tiff(p, img, gt=GT, **kw)
    with host.Raster(p) as r:
        for (x, y, w, h) in [(0, 0, 101, 75), (13, 9, 50, 41), (100, 74, 1, 1), (31, 15, 2, 2), (0, 70, 101, 5),
                             (33, 17, 31, 15)]:
            plan = r.plan(x, y, w, h)
            assert plan is not None and plan[1] == w * h
            want_flags = 1 if kw["compression"] == 1 else (2 if kw.get("predictor") == 2 else 0)
            assert all(c["flags"] == want_flags for c in plan[0])
            assert np.array_equal(_assemble(plan, w, h), img[y:y + h, x:x + w]), (x, y, w, h)
            assert np.array_equal(r.read(x, y, w, h), img[y:y + h, x:x + w])


def test_read_plan_leaves_full_width_raw_strips_of_a_wide_raster_to_the_host_reader(tmp_path):
    """A narrow window of full-width uncompressed strips would stage mostly other blocks' pixels."""
    img = _img(18, 600, 4000)
    p = str(tmp_path / "wide.tif")
    tiffutil.write_tiff(p, img, gt=GT, compression=1, rows_per_strip=600)
    with host.Raster(p) as r:
        assert r.plan(10, 0, 100, 600) is None
        assert r.plan(0, 0, 4000, 600) is not None
        assert np.array_equal(r.read(10, 0, 100, 600), img[:, 10:110])


def test_tiff_writer_extent_writes(tmp_path):
    """gcn10_tiff_put_extent (round 3: a raster's streams of a strip are one extent of the encoder's arena):
    one write per call; decoded pixels equal, with buffered writes and -- where the file system allows it --
    with O_DIRECT (4096-aligned extents)."""
    import ctypes as C
    import zlib
    L = host.lib()
    L.gcn10_tiff_create.restype = C.c_void_p
    L.gcn10_tiff_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_char_p, C.c_size_t]
    L.gcn10_tiff_put_extent.restype = C.c_int
    L.gcn10_tiff_put_extent.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.gcn10_tiff_set_direct.restype = C.c_int
    L.gcn10_tiff_set_direct.argtypes = [C.c_void_p, C.c_bool]
    L.gcn10_tiff_finish.restype = C.c_int
    L.gcn10_tiff_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    W, H = 256 * 5 + 17, 256 * 4 + 3
    rng = np.random.default_rng(5)
    img = np.repeat(np.repeat(rng.integers(0, 200, (H // 32 + 1, W // 32 + 1), dtype=np.uint8), 32, axis=0), 32, axis=1)[:H, :W]
    gt = (C.c_double * 6)(*GT)
    err = C.create_string_buffer(512)
    across, down = (W + 255) // 256, (H + 255) // 256
    for direct in (False, True):
        p = str(tmp_path / ("w%d.tif" % direct))
        w = L.gcn10_tiff_create(p.encode(), W, H, gt, None, err, 512)
        assert w, err.value
        is_direct = direct and L.gcn10_tiff_set_direct(w, True) == 0       # tmpfs refuses: then buffered
        for ty in range(down):                              # one extent per tile row, as the pipeline does per strip
            blobs = []
            for tx in range(across):
                tile = np.zeros((256, 256), np.uint8)
                part = img[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                tile[:part.shape[0], :part.shape[1]] = part
                blobs.append(zlib.compress(tile.tobytes(), 1))
            rel, at = [], 0
            for b in blobs:
                rel.append(at)
                at = (at + len(b) + 15) & ~15
            cap = (at + 4095) & ~4095
            raw = C.create_string_buffer(cap + 4096)
            base = (C.addressof(raw) + 4095) & ~4095        # 4096-aligned, readable to the next multiple
            for b, o in zip(blobs, rel):
                C.memmove(base + o, b, len(b))
            n = len(blobs)
            extent = rel[-1] + len(blobs[-1])
            assert L.gcn10_tiff_put_extent(w, base, extent, n, (C.c_int * n)(*range(across)), (C.c_int * n)(*([ty] * n)),
                                           (C.c_uint32 * n)(*rel), (C.c_uint32 * n)(*[len(b) for b in blobs])) == 0
        assert L.gcn10_tiff_finish(w, err, 512) == 0, err.value
        assert np.array_equal(np.array(Image.open(p)), img), ("direct" if is_direct else "buffered")
    # a stream that reaches past the extent is refused
    w = L.gcn10_tiff_create(str(tmp_path / "bad.tif").encode(), 300, 300, gt, None, err, 512)
    buf = C.create_string_buffer(64)
    assert L.gcn10_tiff_put_extent(w, C.addressof(buf), 32, 1, (C.c_int * 1)(0), (C.c_int * 1)(0), (C.c_uint32 * 1)(16),
                                   (C.c_uint32 * 1)(32)) == -1
    assert L.gcn10_tiff_finish(w, err, 512) != 0


def test_read_plan_of_a_vrt_mosaic(tmp_path):
    img = _img(9, 60, 90)
    img[img == 0] = 7
    tiffutil.write_tiff(str(tmp_path / "a.tif"), img[:, :40], compression=8, tile=(16, 16))
    tiffutil.write_tiff(str(tmp_path / "b.tif"), img[:, 50:], compression=8, rows_per_strip=11)
    tiffutil.write_tiff(str(tmp_path / "c.tif"), img[:, 30:60], compression=8, tile=(16, 16))

    def vrt(sources):
        body = ""
        for name, dx, w, nodata in sources:
            body += ('<ComplexSource><SourceFilename relativeToVRT="1">%s</SourceFilename><SourceBand>1</SourceBand>'
                     '<SrcRect xOff="0" yOff="0" xSize="%d" ySize="60" /><DstRect xOff="%d" yOff="0" xSize="%d" '
                     'ySize="60" />%s</ComplexSource>\n' % (name, w, dx, w, "<NODATA>%d</NODATA>" % nodata
                                                            if nodata is not None else ""))
        (tmp_path / "m.vrt").write_text(
            '<VRTDataset rasterXSize="90" rasterYSize="60">\n<GeoTransform> 0.0, 1.0, 0.0, 0.0, 0.0, -1.0</GeoTransform>\n'
            '<VRTRasterBand dataType="Byte" band="1"><NoDataValue>0</NoDataValue>\n%s</VRTRasterBand></VRTDataset>\n' % body)
        return host.Raster(str(tmp_path / "m.vrt"))

    want = img.copy()
    want[:, 40:50] = 0                                  # the gap between a and b
    with vrt([("a.tif", 0, 40, 0), ("b.tif", 50, 40, 0)]) as r:
        for (x, y, w, h) in [(0, 0, 90, 60), (35, 5, 30, 50), (42, 0, 6, 60), (0, 0, 40, 60)]:
            plan = r.plan(x, y, w, h)
            assert plan is not None
            assert np.array_equal(_assemble(plan, w, h), want[y:y + h, x:x + w])
            assert np.array_equal(r.read(x, y, w, h), want[y:y + h, x:x + w])
        assert r.plan(42, 0, 6, 60)[1] == 0             # nothing covers it: all background
    # sources that overlap inside the window are painted in order by the host reader only
    with vrt([("a.tif", 0, 40, 0), ("c.tif", 30, 30, 0), ("b.tif", 50, 40, 0)]) as r:
        assert r.plan(0, 0, 90, 60) is None
        assert r.plan(0, 0, 30, 60) is not None         # ... where they do not, the plan exists
    # a NODATA other than the background value needs the host reader's transparency
    with vrt([("a.tif", 0, 40, 7)]) as r:
        assert r.plan(0, 0, 40, 60) is None


def test_tiff_writer_gathered_tile_writes(tmp_path):
    """gcn10_tiff_put_tiles (the GPU pipeline's sink: a strip's tiles of one raster per call) writes
    what gcn10_tiff_put_tile would, in any tile order, across its 512-tile batches."""
    import ctypes as C
    import zlib
    L = host.lib()
    L.gcn10_tiff_create.restype = C.c_void_p
    L.gcn10_tiff_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_char_p, C.c_size_t]
    L.gcn10_tiff_put_tiles.restype = C.c_int
    L.gcn10_tiff_put_tiles.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.gcn10_tiff_finish.restype = C.c_int
    L.gcn10_tiff_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    W, H = 256 * 40 + 17, 256 * 15 + 3                  # 41 x 16 = 656 tiles: two batches
    rng = np.random.default_rng(3)
    img = np.repeat(np.repeat(rng.integers(0, 200, (H // 64 + 1, W // 64 + 1), dtype=np.uint8), 64, axis=0), 64, axis=1)[:H, :W]
    gt = (C.c_double * 6)(*GT)
    err = C.create_string_buffer(512)
    p = str(tmp_path / "w.tif")
    w = L.gcn10_tiff_create(p.encode(), W, H, gt, None, err, 512)
    assert w, err.value
    across, down = (W + 255) // 256, (H + 255) // 256
    order = rng.permutation(across * down)
    blobs, txs, tys = [], [], []
    for idx in order:
        ty, tx = divmod(int(idx), across)
        tile = np.zeros((256, 256), np.uint8)
        part = img[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
        tile[:part.shape[0], :part.shape[1]] = part
        blobs.append(zlib.compress(tile.tobytes(), 1))
        txs.append(tx)
        tys.append(ty)
    n = len(blobs)
    bufs = [C.create_string_buffer(b, len(b)) for b in blobs]
    ptrs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p).value for b in bufs])
    sizes = (C.c_uint32 * n)(*[len(b) for b in blobs])
    assert L.gcn10_tiff_put_tiles(w, n, (C.c_int * n)(*txs), (C.c_int * n)(*tys), ptrs, sizes) == 0
    assert L.gcn10_tiff_finish(w, err, 512) == 0, err.value
    Image.MAX_IMAGE_PIXELS = None
    assert np.array_equal(np.array(Image.open(p)), img)
    assert not os.path.exists(p + ".part")
    # a tile outside the raster is refused and fails the file
    w = L.gcn10_tiff_create(p.encode(), 300, 300, gt, None, err, 512)
    assert L.gcn10_tiff_put_tiles(w, 1, (C.c_int * 1)(2), (C.c_int * 1)(0), ptrs, sizes) == -1
    assert L.gcn10_tiff_finish(w, err, 512) != 0


def test_config_lookups_and_conditions_keys(tmp_path):
    """Round-2 keys of the config file (BASELINE config 3, "single lookup"): names in the reference's
    loop order p,f,g x i,ii,iii (src/cn.c:146-147) -> bit k = hc*3 + arc; drained = bit 0 (src/cn.c:145)."""
    base = "hysogs_data_path=a\nesa_data_path=b\nblocks_shp_path=c\nlookup_table_path=d\nlog_dir=e\n"
    p = tmp_path / "c.txt"
    p.write_text(base)
    cfg = host.parse_config(str(p))
    assert cfg["table_mask"] == 0x1FF and cfg["cond_mask"] == 3            # absent = all 18 rasters
    p.write_text(base + "lookups = p_i, g_iii\nconditions=undrained\n")
    cfg = host.parse_config(str(p))
    assert cfg["table_mask"] == (1 << 0) | (1 << 8) and cfg["cond_mask"] == 2
    p.write_text(base + "lookups=all\nconditions=drained,undrained\n")
    cfg = host.parse_config(str(p))
    assert cfg["table_mask"] == 0x1FF and cfg["cond_mask"] == 3
    p.write_text(base + "lookups=g_ii\nlookups=f_i\n")                       # a repeated key: the last one wins
    assert host.parse_config(str(p))["table_mask"] == 1 << 3
    for bad in ("lookups=g_iv\n", "lookups=\n", "conditions=wet\n", "lookups=g-ii\n"):
        p.write_text(base + bad)
        with pytest.raises(host.HostError) as e:
            host.parse_config(str(p))
        assert "bad value for" in str(e.value)

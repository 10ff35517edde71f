"""The gcn10 program end to end: same CLI / config / shapefile / CSV interfaces as the
reference's src/ program, outputs compared (decoded pixels + georeferencing) with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

from oracle import cn_oracle_c as oc
from tests import tiffutil
from tests.conftest import LOOKUPS, ROOT
from tests.util import ESA_NASTY, HSG_NASTY

GCN10 = os.path.join(ROOT, "bin", "gcn10")
CONDS, HCS, ARCS = ("drained", "undrained"), ("p", "f", "g"), ("i", "ii", "iii")

ESA_GT = [10.0, 0.001, 0.0, 50.0, 0.0, -0.001]          # 3000 x 2000 px: lon 10..13, lat 48..50
SOIL_GT = [9.9875, 0.025, 0.0, 50.0125, 0.0, -0.025]    # 25x coarser, origin half a cell off
BLOCKS = [(101, 10.0, 49.0, 11.0, 50.0),     # inside
          (102, 11.0, 48.0, 12.0, 49.0),     # inside
          (103, 12.5, 47.5, 13.5, 48.5),     # sticks out east and south: clamped window
          (104, 20.0, 20.0, 21.0, 21.0)]     # outside the rasters: "invalid raster bounds"


def _world(tmp_path, seed=5, extra_cfg=""):
    rng = np.random.default_rng(seed)
    small = rng.choice(ESA_NASTY, size=(2000 // 20, 3000 // 20))
    esa = np.repeat(np.repeat(small, 20, axis=0), 20, axis=1)
    noise = rng.integers(0, 256, size=esa.shape, dtype=np.uint8)
    esa = np.where(noise < 30, rng.choice(ESA_NASTY, size=esa.shape), esa).astype(np.uint8)
    soil = rng.choice(HSG_NASTY, size=(2000 // 25 + 2, 3000 // 25 + 2)).astype(np.uint8)
    tiffutil.write_tiff(str(tmp_path / "esa.tif"), esa, gt=ESA_GT, compression=8, tile=(512, 512))
    tiffutil.write_tiff(str(tmp_path / "soil_lzw.tif"), soil, gt=SOIL_GT, compression=5, rows_per_strip=8)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), BLOCKS)
    (tmp_path / "config.txt").write_text(
        "# test config\nhysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\n"
        "lookup_table_path=%s\nlog_dir=%s\nstrip_rows=256\nio_threads=4\nworkers_per_gpu=1\n%s"
        % (tmp_path / "soil_lzw.tif", tmp_path / "esa.tif", tmp_path / "blocks.shp", LOOKUPS, tmp_path / "logs",
           extra_cfg))
    return esa, soil


def _run(tmp_path, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([GCN10, *args], cwd=str(tmp_path), capture_output=True, text=True, env=e,
                          timeout=600)


def test_help_and_version_need_nothing():
    out = subprocess.run([GCN10, "--version"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == "gcn10 0.1.0\n"        # src/main.c:51
    out = subprocess.run([GCN10, "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and "--config, -c <file>" in out.stdout and "--overwrite, -o" in out.stdout


def test_missing_config_argument_and_file(tmp_path):
    out = _run(tmp_path)
    assert out.returncode == 1 and "missing -c/--config <file>" in out.stderr     # src/main.c:103-108
    out = _run(tmp_path, "-c", "nope.txt")
    assert out.returncode == 1 and "cannot open config 'nope.txt'" in out.stderr  # src/config.c:52
    (tmp_path / "c.txt").write_text("esa_data_path=x\n")
    out = _run(tmp_path, "-c", "c.txt")
    assert out.returncode == 1 and "missing one of: hysogs_data_path" in out.stderr


def test_no_gpu_means_no_run(tmp_path):
    from gcn10_amd import gpu
    if gpu.device_count() > 0:
        pytest.skip("a GPU is present")
    _world(tmp_path)
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 1
    assert "no CPU fallback" in out.stderr
    assert not (tmp_path / "cn_rasters_drained").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_deflate", [2, 1, 0], ids=["fused-gpu-deflate", "gpu-deflate", "host-zlib"])
def test_blocks_equal_oracle_and_reference_conventions(tmp_path, tables, gpu_deflate):
    esa, soil = _world(tmp_path, extra_cfg="gpu_deflate=%d\n" % gpu_deflate)
    (tmp_path / "ids.txt").write_text("101 102\n103\n104 999\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert "processing 5 blocks from list file" in log                           # src/main.c:165
    assert "[ERROR] [rank 0] invalid raster bounds for" in log                    # block 104, src/raster.c:143
    assert "[ERROR] [rank 0] esa load failed for block 104" in log                # src/cn.c:189
    assert "[ERROR] [rank 0] block 999 not found" in log                          # src/cn.c:173
    assert "processed 5 blocks on 1 ranks" in log                                 # src/main.c:191
    assert re.search(r"timing gpu 0 \(pci [0-9a-f:.?]+, numa node -?\d+\): 1 worker\(s\), \d+ blocks, worker seconds: "
                     r"in blocks [\d.]+, reading landcover [\d.]+, waiting for gpu [\d.]+, waiting for sink", log)
    assert len(re.findall(r"completed condition for 101: ", log)) == 18           # src/cn.c:367
    assert len(re.findall(r"progress: completed block 101 / total 5", log)) == 18  # src/log.c:203, 18x
    assert "completed condition for 101: drained/p/i" in log and \
        "completed condition for 101: undrained/g/iii" in log

    for bid, *bbox in BLOCKS[:3]:
        xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
        sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
        want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx],
                                    sgt, tables)
        if bid == 103:
            assert (W, H) == (500, 500)           # clamped at the east / south edge
        for c, cond in enumerate(CONDS):
            for hi, hc in enumerate(HCS):
                for ai, arc in enumerate(ARCS):
                    p = tmp_path / ("cn_rasters_%s" % cond) / ("cn_%s_%s_%d.tif" % (hc, arc, bid))  # src/cn.c:308
                    im = Image.open(str(p))
                    assert np.array_equal(np.array(im), want[c * 9 + hi * 3 + ai]), p
                    t = im.tag_v2
                    assert t[259] == 8 and t[322] == 256 and t[323] == 256        # src/raster.c:206-207
                    assert tuple(t[33550]) == (gt[1], -gt[5], 0.0)                # clipped ESA gt, src/raster.c:157-162
                    assert tuple(t[33922]) == (0.0, 0.0, 0.0, gt[0], gt[3], 0.0)
                    assert 4326 in t[34735]
    assert not (tmp_path / "cn_rasters_drained" / "cn_p_i_104.tif").exists()
    assert sorted(os.listdir(tmp_path / "cn_rasters_undrained")) == sorted(
        "cn_%s_%s_%d.tif" % (hc, arc, b) for hc in HCS for arc in ARCS for b in (101, 102, 103))


@pytest.mark.gpu
def test_overwrite_rule_and_shapefile_mode(tmp_path, tables):
    esa, soil = _world(tmp_path, seed=9)
    (tmp_path / "ids.txt").write_text("102\n")
    assert _run(tmp_path, "-c", "config.txt", "-b", "ids.txt").returncode == 0      # -b as advertised, src/main.c:28
    first = (tmp_path / "cn_rasters_drained" / "cn_f_i_102.tif").read_bytes()
    # second run without -o: existing outputs stay, new ones get a trailing underscore (src/cn.c:320-360)
    assert _run(tmp_path, "--config", "config.txt", "--blocks", "ids.txt").returncode == 0
    assert (tmp_path / "cn_rasters_drained" / "cn_f_i_102.tif").read_bytes() == first
    again = tmp_path / "cn_rasters_drained" / "cn_f_i_102_.tif"
    assert again.exists() and np.array_equal(np.array(Image.open(str(again))),
                                             np.array(Image.open(str(tmp_path / "cn_rasters_drained" / "cn_f_i_102.tif"))))
    # with -o nothing new appears
    n_before = len(os.listdir(tmp_path / "cn_rasters_drained"))
    assert _run(tmp_path, "-c", "config.txt", "-l", "ids.txt", "-o").returncode == 0
    assert len(os.listdir(tmp_path / "cn_rasters_drained")) == n_before
    # no list: every ID of the shapefile (the mode that crashes in the reference)
    out = _run(tmp_path, "-c", "config.txt", "-o")
    assert out.returncode == 0
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert "processing 4 blocks from shapefile" in log
    assert (tmp_path / "cn_rasters_undrained" / "cn_g_iii_103.tif").exists()


@pytest.mark.gpu
def test_default_is_two_workers_per_gpu(tmp_path):
    _world(tmp_path, seed=23)
    cfg = (tmp_path / "config.txt").read_text().replace("workers_per_gpu=1\n", "")
    (tmp_path / "config.txt").write_text(cfg)
    out = _run(tmp_path, "-c", "config.txt", "--gpus", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    log0 = (tmp_path / "logs" / "rank_0.log").read_text()
    assert "starting processing with 2 gpu workers" in log0 and "processed 4 blocks on 2 ranks" in log0
    assert (tmp_path / "logs" / "rank_1.log").exists()
    assert len(os.listdir(tmp_path / "cn_rasters_drained")) == 27


@pytest.mark.gpu
def test_broken_lookup_aborts_like_the_reference(tmp_path):
    _world(tmp_path)
    (tmp_path / "lk").mkdir()
    for f in os.listdir(LOOKUPS):
        if f != "default_lookup_g_ii.csv":
            (tmp_path / "lk" / f).write_bytes(open(os.path.join(LOOKUPS, f), "rb").read())
    cfg = (tmp_path / "config.txt").read_text().replace(LOOKUPS, str(tmp_path / "lk"))
    (tmp_path / "config.txt").write_text(cfg)
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 1                                                    # MPI_Abort(.., 1), src/cn.c:32
    assert "cannot open lookup table" in out.stderr and "default_lookup_g_ii.csv" in out.stderr


@pytest.mark.gpu
def test_null_sink_runs_the_gpu_pipeline_without_files(tmp_path):
    _world(tmp_path)
    out = _run(tmp_path, "-c", "config.txt", env={"GCN10_SINK": "null"})
    assert out.returncode == 0
    assert not (tmp_path / "cn_rasters_drained").exists()
    assert "completed condition for 101: drained/p/i" in (tmp_path / "logs" / "rank_0.log").read_text()


@pytest.mark.gpu
def test_several_workers_share_the_block_queue(tmp_path, tables):
    """Three worker threads ("ranks") on the one GPU of the test box pull from the same
    atomic block counter; every block is produced exactly once, bit-identical."""
    esa, soil = _world(tmp_path, seed=21)
    out = _run(tmp_path, "-c", "config.txt", "--gpus", "3", env={"GCN10_OVERSUBSCRIBE": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    logs = [(tmp_path / "logs" / ("rank_%d.log" % i)).read_text() for i in range(3)]
    assert "starting processing with 3 gpu workers" in logs[0]
    started = sum(len(re.findall(r"processing block \d+", l)) for l in logs)
    assert started == 4                                           # each block taken once
    for bid, *bbox in BLOCKS[:3]:
        xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
        sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
        want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx],
                                    sgt, tables)
        for r in (0, 7, 17):
            c, k = divmod(r, 9)
            p = tmp_path / ("cn_rasters_%s" % CONDS[c]) / ("cn_%s_%s_%d.tif" % (HCS[k // 3], ARCS[k % 3], bid))
            assert np.array_equal(np.array(Image.open(str(p))), want[r])


@pytest.mark.gpu
def test_outer_launcher_ranks_take_round_robin_shares(tmp_path):
    """Started by mpirun / srun, process r of n takes blocks r, r+n, ... (src/main.c:171)."""
    _world(tmp_path, seed=22)
    (tmp_path / "ids.txt").write_text("101 102 103\n")
    # the two processes of the "launch" run side by side, as under mpirun; they end with the reference's closing
    # barrier (src/main.c:187-194) through marker files in the log directory, and rank 0 logs the totals
    env = dict(os.environ, PMI_SIZE="2", GCN10_JOB_ID="t22", GCN10_BARRIER_SECONDS="300")
    procs = [subprocess.Popen([GCN10, "-c", "config.txt", "-l", "ids.txt"], cwd=str(tmp_path), text=True,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, PMI_RANK=str(rank)))
             for rank in (1, 0)]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    log0 = (tmp_path / "logs" / "rank_0.log").read_text()
    log1 = (tmp_path / "logs" / "rank_1.log").read_text()
    assert "process 0 of 2: 2 of 3 blocks" in log0 and "process 1 of 2: 1 of 3 blocks" in log1
    assert "processing block 101" in log0 and "processing block 103" in log0 and "processing block 102" in log1
    assert "all 2 processes: processed 3 blocks on 2 ranks" in log0 and "all 2 processes" not in log1
    assert not [f for f in os.listdir(tmp_path / "logs") if f.startswith(".done_")]
    assert len(os.listdir(tmp_path / "cn_rasters_drained")) == 27


@pytest.mark.gpu
def test_vrt_landcover_with_local_tile_mirror(tmp_path, tables):
    """esa_data_path is a VRT of /vsicurl/ tiles, as in the reference's src/test/config.txt; offline
    its sources resolve to esa_tile_dir.  Two tiles side by side, a block straddling both."""
    rng = np.random.default_rng(31)
    esa = rng.choice(ESA_NASTY, size=(2000, 3000)).astype(np.uint8)
    esa[esa == 0] = 10                                  # ComplexSource NODATA=0 stays transparent
    soil = rng.choice(HSG_NASTY, size=(82, 122)).astype(np.uint8)
    (tmp_path / "tiles").mkdir()
    tiffutil.write_tiff(str(tmp_path / "tiles" / "T_W.tif"), esa[:, :1500], compression=8, tile=(512, 512))
    tiffutil.write_tiff(str(tmp_path / "tiles" / "T_E.tif"), esa[:, 1500:], compression=5, tile=(256, 256))
    src = ""
    for name, dx in (("T_W.tif", 0), ("T_E.tif", 1500)):
        src += ('<ComplexSource resampling="nearest"><SourceFilename relativeToVRT="0">/vsicurl/https://example.invalid/'
                'map/%s</SourceFilename><SourceBand>1</SourceBand><SrcRect xOff="0" yOff="0" xSize="1500" ySize="2000" />'
                '<DstRect xOff="%d" yOff="0" xSize="1500" ySize="2000" /><NODATA>0</NODATA></ComplexSource>\n' % (name, dx))
    (tmp_path / "esa.vrt").write_text(
        '<VRTDataset rasterXSize="3000" rasterYSize="2000">\n<GeoTransform> %r, %r, 0.0, %r, 0.0, %r</GeoTransform>\n'
        '<VRTRasterBand dataType="Byte" band="1"><NoDataValue>0</NoDataValue>\n%s</VRTRasterBand></VRTDataset>\n'
        % (ESA_GT[0], ESA_GT[1], ESA_GT[3], ESA_GT[5], src))
    tiffutil.write_tiff(str(tmp_path / "soil.tif"), soil, gt=SOIL_GT, compression=5, rows_per_strip=8)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), [(7, 11.0, 48.5, 12.0, 49.5)])   # columns 1000..2000
    (tmp_path / "config.txt").write_text(
        "hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
        "esa_tile_dir=%s\nstrip_rows=512\n" % (tmp_path / "soil.tif", tmp_path / "esa.vrt", tmp_path / "blocks.shp",
                                               LOOKUPS, tmp_path / "logs", tmp_path / "tiles"))
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    bbox = [11.0, 48.5, 12.0, 49.5]
    xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
    sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
    assert xo < 1500 < xo + W                           # the window spans both tiles
    want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
    for r in (0, 4, 8, 9, 13, 17):
        c, k = divmod(r, 9)
        p = tmp_path / ("cn_rasters_%s" % CONDS[c]) / ("cn_%s_%s_7.tif" % (HCS[k // 3], ARCS[k % 3]))
        im = Image.open(str(p))
        assert np.array_equal(np.array(im), want[r])
        assert 4326 in im.tag_v2[34735]                 # VRT input: WGS84 default GeoKeys
    # without the mirror the landcover cannot be read: the block is skipped with the reference's message
    cfg = (tmp_path / "config.txt").read_text().replace("esa_tile_dir=%s\n" % (tmp_path / "tiles"), "")
    (tmp_path / "config.txt").write_text(cfg)
    out = _run(tmp_path, "-c", "config.txt", "-o")
    assert out.returncode == 0
    assert "esa load failed for block 7" in (tmp_path / "logs" / "rank_0.log").read_text()


@pytest.mark.gpu
@pytest.mark.parametrize("buffers", ["2", "4"])
def test_compressed_tiles_spill_past_the_pinned_arena(tmp_path, tables, buffers):
    """The pinned arena holds an eighth of the encoder's worst case; a strip that needs more goes
    through a pageable buffer.  Forced here with a 64 KB arena: results must not change.  Also the
    two other strip-buffer counts (the default is three)."""
    esa, soil = _world(tmp_path, seed=41)
    (tmp_path / "ids.txt").write_text("101\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt",
               env={"GCN10_PINNED_ARENA_BYTES": "65536", "GCN10_STRIP_BUFFERS": buffers})
    assert out.returncode == 0, out.stderr[-2000:]
    bid, *bbox = BLOCKS[0]
    xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
    sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
    want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
    for r in range(18):
        c, k = divmod(r, 9)
        p = tmp_path / ("cn_rasters_%s" % CONDS[c]) / ("cn_%s_%s_%d.tif" % (HCS[k // 3], ARCS[k % 3], bid))
        assert np.array_equal(np.array(Image.open(str(p))), want[r])


def _check_block(tmp_path, esa, soil, tables, bid, bbox, esa_gt=ESA_GT, rasters=range(18)):
    xo, yo, W, H, gt = oc.window(esa_gt, esa.shape[1], esa.shape[0], bbox)
    sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
    want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
    for r in rasters:
        c, k = divmod(r, 9)
        p = tmp_path / ("cn_rasters_%s" % CONDS[c]) / ("cn_%s_%s_%d.tif" % (HCS[k // 3], ARCS[k % 3], bid))
        assert np.array_equal(np.array(Image.open(str(p))), want[r]), p


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_inflate", [1, 0], ids=["gpu-inflate", "host-inflate"])
@pytest.mark.parametrize("layout", ["tiles", "strips"])
def test_deflate_landcover_decoded_on_the_gpu_or_the_host(tmp_path, tables, gpu_inflate, layout):
    """DEFLATE landcover (tiles as in the ESA files, or strips): the compressed chunks go to the GPU
    and are decoded there (gpu_inflate=1, default) or on the I/O pool (0); same rasters either way."""
    esa, soil = _world(tmp_path, seed=77, extra_cfg="gpu_inflate=%d\n" % gpu_inflate)
    if layout == "strips":
        tiffutil.write_tiff(str(tmp_path / "esa.tif"), esa, gt=ESA_GT, compression=8, rows_per_strip=37)
    (tmp_path / "ids.txt").write_text("101 103\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert ("gpu inflate of deflate landcover" in log) == bool(gpu_inflate)
    for bid, *bbox in (BLOCKS[0], BLOCKS[2]):
        _check_block(tmp_path, esa, soil, tables, bid, bbox)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["raw-tiles", "raw-strips", "deflate-predictor2-tiles", "deflate-predictor2-strips",
                                  "lzw-predictor2-tiles"])
@pytest.mark.parametrize("prefetch", [1, 0], ids=["one-block-ahead", "in-turn"])
def test_raw_and_predictor2_landcover_through_the_gpu_side(tmp_path, tables, kind, prefetch):
    """Round 3: uncompressed landcover (the literal path of the north star: raw bytes -> pinned ring ->
    hipMemcpyAsync -> HBM, untiled there) and DEFLATE landcover written with TIFF predictor 2 (summed back on
    the GPU) give the oracle's rasters; LZW stays with the host reader and gives them too.  Both with the
    input thread one block ahead of the encoder and with input and encode in turn."""
    esa, soil = _world(tmp_path, seed=81, extra_cfg="prefetch_blocks=%d\n" % prefetch)
    kw = {"raw-tiles": dict(compression=1, tile=(256, 128)), "raw-strips": dict(compression=1, rows_per_strip=16),
          "deflate-predictor2-tiles": dict(compression=8, predictor=2, tile=(512, 512)),
          "deflate-predictor2-strips": dict(compression=8, predictor=2, rows_per_strip=37),
          "lzw-predictor2-tiles": dict(compression=5, predictor=2, tile=(256, 256))}[kind]
    tiffutil.write_tiff(str(tmp_path / "esa.tif"), esa, gt=ESA_GT, **kw)
    (tmp_path / "ids.txt").write_text("101 102 103\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert ("one block ahead of the encoder" in log) == bool(prefetch)
    for bid, *bbox in BLOCKS[:3]:
        _check_block(tmp_path, esa, soil, tables, bid, bbox, rasters=(0, 4, 8, 9, 13, 17))


@pytest.mark.gpu
@pytest.mark.parametrize("predictor", [1, 2])
def test_vrt_of_cog_shaped_landcover_tiles(tmp_path, tables, predictor):
    """The landcover as the shipped VRT names it (/root/reference/landcover/esa_worldcover_2021.vrt:265-272): a
    mosaic of cloud-optimised GeoTIFFs -- full-resolution IFD first, two overview IFDs behind it, tile data laid
    out overviews first, 1024 x 1024 DEFLATE blocks, with and without predictor 2 -- resolved through
    esa_tile_dir; the block's window crosses the seam between the two files.  The reader must take IFD 0's
    tiles, the GPU plan must accept them."""
    from gcn10_amd import host
    rng = np.random.default_rng(41)
    small = rng.choice(ESA_NASTY, size=(2000 // 16, 3000 // 16))
    esa = np.repeat(np.repeat(small, 16, axis=0), 16, axis=1)
    esa = np.where(rng.random(esa.shape) < 0.1, rng.choice(ESA_NASTY, size=esa.shape), esa).astype(np.uint8)
    esa[esa == 0] = 10
    soil = rng.choice(HSG_NASTY, size=(82, 122)).astype(np.uint8)
    (tmp_path / "tiles").mkdir()
    tiffutil.write_cog(str(tmp_path / "tiles" / "C_W.tif"), esa[:, :1536], tile=(1024, 1024), overviews=2, predictor=predictor)
    tiffutil.write_cog(str(tmp_path / "tiles" / "C_E.tif"), esa[:, 1536:], tile=(1024, 1024), overviews=2, predictor=predictor)
    src = ""
    for name, dx, w in (("C_W.tif", 0, 1536), ("C_E.tif", 1536, 1464)):
        src += ('<ComplexSource resampling="nearest"><SourceFilename relativeToVRT="0">/vsicurl/https://example.invalid/'
                'map/%s</SourceFilename><SourceBand>1</SourceBand><SrcRect xOff="0" yOff="0" xSize="%d" ySize="2000" />'
                '<DstRect xOff="%d" yOff="0" xSize="%d" ySize="2000" /><NODATA>0</NODATA></ComplexSource>\n'
                % (name, w, dx, w))
    (tmp_path / "esa.vrt").write_text(
        '<VRTDataset rasterXSize="3000" rasterYSize="2000">\n<GeoTransform> %r, %r, 0.0, %r, 0.0, %r</GeoTransform>\n'
        '<VRTRasterBand dataType="Byte" band="1"><NoDataValue>0</NoDataValue>\n%s</VRTRasterBand></VRTDataset>\n'
        % (ESA_GT[0], ESA_GT[1], ESA_GT[3], ESA_GT[5], src))
    # the plan exists (the GPU side takes these files) and names full-resolution chunks only
    with host.Raster(str(tmp_path / "esa.vrt"), str(tmp_path / "tiles")) as r:
        plan = r.plan(1000, 500, 1000, 1000)
        assert plan is not None and plan[1] == 1000 * 1000
        assert all(c["chunk_w"] == 1024 and c["flags"] == (2 if predictor == 2 else 0) for c in plan[0])
        assert np.array_equal(r.read(1000, 500, 1000, 1000), esa[500:1500, 1000:2000])
    tiffutil.write_tiff(str(tmp_path / "soil.tif"), soil, gt=SOIL_GT, compression=5, rows_per_strip=8)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), [(7, 11.0, 48.5, 12.0, 49.5)])
    (tmp_path / "config.txt").write_text(
        "hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
        "esa_tile_dir=%s\nstrip_rows=512\n" % (tmp_path / "soil.tif", tmp_path / "esa.vrt", tmp_path / "blocks.shp",
                                               LOOKUPS, tmp_path / "logs", tmp_path / "tiles"))
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    _check_block(tmp_path, esa, soil, tables, 7, [11.0, 48.5, 12.0, 49.5], rasters=(0, 5, 9, 17))


@pytest.mark.gpu
@pytest.mark.parametrize("direct", [0, 1])
def test_one_extent_per_raster_and_strip_with_and_without_direct_io(tmp_path, tables, direct):
    """Round 3: the encoder lays a raster's streams of a strip out as one extent (pass B'), the sink appends it
    with one write -- buffered, or with O_DIRECT (direct_io=1) where the file system takes it (it falls back
    where it does not, e.g. tmpfs).  Same decoded rasters, same tags."""
    esa, soil = _world(tmp_path, seed=55, extra_cfg="direct_io=%d\n" % direct)
    (tmp_path / "ids.txt").write_text("101 103\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    for bid, *bbox in (BLOCKS[0], BLOCKS[2]):
        _check_block(tmp_path, esa, soil, tables, bid, bbox)


@pytest.mark.gpu
def test_vrt_of_deflate_tiles_with_a_gap_decodes_on_the_gpu(tmp_path, tables):
    """A mosaic of two DEFLATE tiles that leaves a strip of the block uncovered: the chunks of both
    files are decoded on the GPU, the gap reads as 0 (the VRT's NoDataValue)."""
    rng = np.random.default_rng(32)
    esa = rng.choice(ESA_NASTY, size=(2000, 3000)).astype(np.uint8)
    esa[esa == 0] = 10
    esa[:, 1400:1500] = 0                               # nothing covers these columns
    soil = rng.choice(HSG_NASTY, size=(82, 122)).astype(np.uint8)
    (tmp_path / "tiles").mkdir()
    tiffutil.write_tiff(str(tmp_path / "tiles" / "T_W.tif"), esa[:, :1400], compression=8, tile=(512, 512))
    tiffutil.write_tiff(str(tmp_path / "tiles" / "T_E.tif"), esa[:, 1500:], compression=8, tile=(256, 128))
    src = ""
    for name, dx, w in (("T_W.tif", 0, 1400), ("T_E.tif", 1500, 1500)):
        src += ('<ComplexSource resampling="nearest"><SourceFilename relativeToVRT="0">/vsicurl/https://example.invalid/'
                'map/%s</SourceFilename><SourceBand>1</SourceBand><SrcRect xOff="0" yOff="0" xSize="%d" ySize="2000" />'
                '<DstRect xOff="%d" yOff="0" xSize="%d" ySize="2000" /><NODATA>0</NODATA></ComplexSource>\n'
                % (name, w, dx, w))
    (tmp_path / "esa.vrt").write_text(
        '<VRTDataset rasterXSize="3000" rasterYSize="2000">\n<GeoTransform> %r, %r, 0.0, %r, 0.0, %r</GeoTransform>\n'
        '<VRTRasterBand dataType="Byte" band="1"><NoDataValue>0</NoDataValue>\n%s</VRTRasterBand></VRTDataset>\n'
        % (ESA_GT[0], ESA_GT[1], ESA_GT[3], ESA_GT[5], src))
    tiffutil.write_tiff(str(tmp_path / "soil.tif"), soil, gt=SOIL_GT, compression=5, rows_per_strip=8)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), [(7, 11.0, 48.5, 12.0, 49.5)])
    (tmp_path / "config.txt").write_text(
        "hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
        "esa_tile_dir=%s\nstrip_rows=512\n" % (tmp_path / "soil.tif", tmp_path / "esa.vrt", tmp_path / "blocks.shp",
                                               LOOKUPS, tmp_path / "logs", tmp_path / "tiles"))
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    _check_block(tmp_path, esa, soil, tables, 7, [11.0, 48.5, 12.0, 49.5], rasters=(0, 5, 9, 17))


@pytest.mark.gpu
def test_corrupt_landcover_tile_fails_its_block_only(tmp_path, tables):
    """A tile whose DEFLATE stream is damaged: the block that needs it is skipped with the
    reference's load_raster failure lines and leaves no files; other blocks are written."""
    esa, soil = _world(tmp_path, seed=78)
    path = tmp_path / "esa.tif"
    raw = bytearray(path.read_bytes())
    im = Image.open(str(path))
    offs, cnts = im.tag_v2[324], im.tag_v2[325]
    across = (3000 + 511) // 512
    k = 2 * across + 3                                  # tile (row 2, col 3): rows 1024.., columns 1536..: block 102 only
    for i in range(offs[k] + 2, offs[k] + cnts[k]):
        raw[i] = 0xFF
    path.write_bytes(bytes(raw))
    (tmp_path / "ids.txt").write_text("101 102\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert "gdalrasterio error: cannot decode a tile" in log
    assert "esa load failed for block 102" in log
    assert not (tmp_path / "cn_rasters_drained" / "cn_p_i_102.tif").exists()
    _check_block(tmp_path, esa, soil, tables, 101, BLOCKS[0][1:])


@pytest.mark.gpu
def test_custom_lookups_with_more_than_256_pixel_classes(tmp_path):
    """User-supplied lookup CSVs whose 18 rasters distinguish more than 256 (landcover, soil) pairs:
    the fused encoder's class map does not exist, the program says so and encodes per raster."""
    from oracle import cn_oracle_np as onp
    esa, soil = _world(tmp_path, seed=91)
    rng = np.random.default_rng(92)
    lk = tmp_path / "lookups"
    lk.mkdir()
    tabs = []
    for hc in onp.HCS:
        for arc in onp.ARCS:
            lines = ["grid_code,cn"]
            for lc in sorted(set(range(64)) | set(int(v) for v in ESA_NASTY)):      # 64+ live classes x 8 soil pairs
                for g in "ABCD":
                    lines.append("%d_%s,%d" % (lc, g, int(rng.integers(0, 255))))
            p = lk / ("default_lookup_%s_%s.csv" % (hc, arc))
            p.write_text("\n".join(lines) + "\n")
            t, bad = oc.load_lookup_table(str(p))
            assert bad == 0
            tabs.append(t)
    tabs = np.stack(tabs)
    cfg = (tmp_path / "config.txt").read_text().replace("lookup_table_path=%s" % LOOKUPS, "lookup_table_path=%s" % lk)
    (tmp_path / "config.txt").write_text(cfg)
    (tmp_path / "ids.txt").write_text("101\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert "more than 256 pixel classes" in log
    _check_block(tmp_path, esa, soil, tabs, 101, BLOCKS[0][1:])


@pytest.mark.gpu
def test_full_size_block_of_the_real_vrt_shape(tmp_path, tables):
    """One block as the shipped VRT cuts it: 36001 x 36001 pixels (pixel size 8.333...e-05, SURVEY
    section 7), rows not 16-byte aligned, DEFLATE landcover in 1024 x 1024 tiles.  The whole
    program runs on it; three of the 18 rasters are decoded by libtiff and compared with the
    oracle on 600 rows."""
    import bench
    Image.MAX_IMAGE_PIXELS = None
    size, px = 36001, 8.3333333333330430e-05
    esa, _, coarse, _ = bench.synth_block(5, size, "patches")
    hs = coarse.shape[0]
    egt = [0.0, px, 0.0, 3.0, 0.0, -px]
    sgt = [0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs]
    tiffutil.write_tiff(str(tmp_path / "esa.tif"), esa, gt=egt, compression=8, tile=(1024, 1024))
    tiffutil.write_tiff(str(tmp_path / "soil.tif"), coarse, gt=sgt, compression=5, rows_per_strip=16)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), [(1, 0.0, 0.0, 3.0, 3.0)])
    (tmp_path / "config.txt").write_text(
        "hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
        % (tmp_path / "soil.tif", tmp_path / "esa.tif", tmp_path / "blocks.shp", LOOKUPS, tmp_path / "logs"))
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    xo, yo, W, H, gt = oc.window(egt, size, size, [0.0, 0.0, 3.0, 3.0])
    assert (W, H) == (size, size)
    sxo, syo, hsx, hsy, sg = oc.window(sgt, hs, hs, [0.0, 0.0, 3.0, 3.0])
    y0 = 20000
    want = oc.process_block_mem(esa[yo + y0:yo + y0 + 600, xo:xo + W], [gt[0], gt[1], 0.0, gt[3] + y0 * gt[5], 0.0, gt[5]],
                                coarse[syo:syo + hsy, sxo:sxo + hsx], sg, tables)
    for r in (0, 13, 17):
        c, k = divmod(r, 9)
        p = tmp_path / ("cn_rasters_%s" % CONDS[c]) / ("cn_%s_%s_1.tif" % (HCS[k // 3], ARCS[k % 3]))
        im = np.array(Image.open(str(p)))
        assert im.shape == (size, size)
        assert np.array_equal(im[y0:y0 + 600], want[r]), p


# ---- BASELINE config 3: "single lookup" -- a subset of the 18 rasters ---------------------------

def test_lookups_and_conditions_values_are_checked(tmp_path):
    _world(tmp_path, extra_cfg="lookups=g_iv\n")
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 1 and "bad value for lookups: 'g_iv'" in out.stderr
    _world(tmp_path, extra_cfg="conditions=dry\n")
    out = _run(tmp_path, "-c", "config.txt")
    assert out.returncode == 1 and "bad value for conditions" in out.stderr
    _world(tmp_path)
    out = _run(tmp_path, "-c", "config.txt", "--lookups", "q_i")
    assert out.returncode == 1 and "bad --lookups / --conditions value" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_deflate", [2, 1, 0], ids=["fused-gpu-deflate", "gpu-deflate", "host-zlib"])
@pytest.mark.parametrize("how", ["config", "flags"])
def test_single_lookup_run_equals_oracle_subset(tmp_path, tables, gpu_deflate, how):
    """lookups=g_ii conditions=drained (config keys or --lookups / --conditions): one raster per block,
    equal to the oracle's subset pass; nothing else is written, logged or even read (the other eight
    CSVs need not exist)."""
    import shutil
    only = tmp_path / "lookups_only_g_ii"
    only.mkdir()
    shutil.copy(os.path.join(LOOKUPS, "default_lookup_g_ii.csv"), only / "default_lookup_g_ii.csv")
    extra = "gpu_deflate=%d\n" % gpu_deflate
    if how == "config":
        extra += "lookups = g_ii\nconditions=drained\n"
    esa, soil = _world(tmp_path, extra_cfg=extra)
    cfg = (tmp_path / "config.txt").read_text().replace("lookup_table_path=%s" % LOOKUPS, "lookup_table_path=%s" % only)
    (tmp_path / "config.txt").write_text(cfg)
    (tmp_path / "ids.txt").write_text("101 103\n")
    flags = [] if how == "config" else ["--lookups", "g_ii", "--conditions", "drained"]
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt", *flags)
    assert out.returncode == 0, out.stderr[-2000:]
    log = (tmp_path / "logs" / "rank_0.log").read_text()
    assert re.findall(r"completed condition for 101: (\S+)", log) == ["drained/g/ii"]
    assert len(re.findall(r"progress: completed block 103 / total 2", log)) == 1
    assert not (tmp_path / "cn_rasters_undrained").exists()
    assert sorted(os.listdir(tmp_path / "cn_rasters_drained")) == ["cn_g_ii_101.tif", "cn_g_ii_103.tif"]
    k = 2 * 3 + 1                                   # hc = g, arc = ii
    for bid, *bbox in (BLOCKS[0], BLOCKS[2]):
        xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
        sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
        want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt,
                                    tables, cond_mask=1, table_mask=1 << k)
        got = np.array(Image.open(str(tmp_path / "cn_rasters_drained" / ("cn_g_ii_%d.tif" % bid))))
        assert np.array_equal(got, want[k]), bid


@pytest.mark.gpu
def test_lookup_subset_of_several_tables_and_both_conditions(tmp_path, tables):
    esa, soil = _world(tmp_path, extra_cfg="lookups=p_i, f_iii,g_ii\n")
    (tmp_path / "ids.txt").write_text("102\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt", "--conditions", "undrained,drained")
    assert out.returncode == 0, out.stderr[-2000:]
    names = ["cn_%s_102.tif" % n for n in ("f_iii", "g_ii", "p_i")]
    assert sorted(os.listdir(tmp_path / "cn_rasters_drained")) == names
    assert sorted(os.listdir(tmp_path / "cn_rasters_undrained")) == names
    bid, *bbox = BLOCKS[1]
    xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
    sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
    mask = (1 << 0) | (1 << 5) | (1 << 7)
    want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables,
                                table_mask=mask)
    for c, cond in enumerate(CONDS):
        for k, name in ((0, "p_i"), (5, "f_iii"), (7, "g_ii")):
            got = np.array(Image.open(str(tmp_path / ("cn_rasters_%s" % cond) / ("cn_%s_102.tif" % name))))
            assert np.array_equal(got, want[c * 9 + k]), (cond, name)


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_deflate", [1, 0], ids=["gpu-deflate", "host-zlib"])
def test_wide_block_through_the_vector_strip_kernels(tmp_path, tables, gpu_deflate):
    """A block 1237 px wide (odd: rows not 16-byte aligned) in the two modes that materialise CN strips:
    the 16-byte-per-lane strip kernels run (blocks under 1040 px take the byte-wise kernel), strips of
    256 rows start at multiples of 16 bytes, lanes straddle row ends."""
    esa, soil = _world(tmp_path, seed=21, extra_cfg="gpu_deflate=%d\n" % gpu_deflate)
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), [(301, 10.5, 48.9, 11.737, 49.9)])
    (tmp_path / "ids.txt").write_text("301\n")
    out = _run(tmp_path, "-c", "config.txt", "-l", "ids.txt")
    assert out.returncode == 0, out.stderr[-2000:]
    bbox = [10.5, 48.9, 11.737, 49.9]
    xo, yo, W, H, gt = oc.window(ESA_GT, 3000, 2000, bbox)
    assert W == 1237 and H == 1000
    sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
    want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
    for c, cond in enumerate(CONDS):
        for hi, hc in enumerate(HCS):
            for ai, arc in enumerate(ARCS):
                p = tmp_path / ("cn_rasters_%s" % cond) / ("cn_%s_%s_301.tif" % (hc, arc))
                assert np.array_equal(np.array(Image.open(str(p))), want[c * 9 + hi * 3 + ai]), p

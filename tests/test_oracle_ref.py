"""Pins the oracle against the reference itself -- on a machine where `make -C oracle ref` could
build oracle/_ref/gcn10_ref (needs GDAL and MPI; this image has neither GDAL nor, on the GPU box,
/root/reference, so here the test skips and parity stays "unpinned", oracle/README.md).

The reference program is run as its README runs it (mpirun -n 1 gcn10 -c config -l list -o) on a
small synthetic world written by tests/tiffutil.py; its 18 GeoTIFFs per block are read back and
compared with the oracle's rasters."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import tiffutil
from tests.conftest import LOOKUPS, ROOT
from tests.util import ESA_NASTY, HSG_NASTY

REF = os.path.join(ROOT, "oracle", "_ref", "gcn10_ref")

pytestmark = pytest.mark.skipif(not os.path.exists(REF) or shutil.which("mpirun") is None,
                                reason="oracle/_ref/gcn10_ref not built (no GDAL in this image): parity unpinned")


def test_reference_program_equals_oracle(tmp_path, tables):
    from PIL import Image
    from oracle import cn_oracle_c as oc
    rng = np.random.default_rng(3)
    esa_gt = [10.0, 0.001, 0.0, 50.0, 0.0, -0.001]
    soil_gt = [9.9875, 0.025, 0.0, 50.0125, 0.0, -0.025]
    small = rng.choice(ESA_NASTY, size=(50, 75))
    esa = np.repeat(np.repeat(small, 20, axis=0), 20, axis=1).astype(np.uint8)
    soil = rng.choice(HSG_NASTY, size=(42, 62)).astype(np.uint8)
    tiffutil.write_tiff(str(tmp_path / "esa.tif"), esa, gt=esa_gt, compression=8, tile=(256, 256))
    tiffutil.write_tiff(str(tmp_path / "soil.tif"), soil, gt=soil_gt, compression=5, rows_per_strip=8)
    blocks = [(1, 10.0, 49.5, 10.5, 50.0), (2, 10.7, 49.0, 11.6, 49.6)]       # the second sticks out east
    tiffutil.write_block_shapefile(str(tmp_path / "blocks"), blocks)
    (tmp_path / "config.txt").write_text(
        "hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
        % (tmp_path / "soil.tif", tmp_path / "esa.tif", tmp_path / "blocks.shp", LOOKUPS, tmp_path / "logs"))
    (tmp_path / "ids.txt").write_text("1 2\n")
    out = subprocess.run(["mpirun", "-n", "1", REF, "-c", "config.txt", "-l", "ids.txt", "-o"], cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    conds, hcs, arcs = ("drained", "undrained"), ("p", "f", "g"), ("i", "ii", "iii")
    for bid, *bbox in blocks:
        xo, yo, W, H, gt = oc.window(esa_gt, esa.shape[1], esa.shape[0], bbox)
        sxo, syo, hsx, hsy, sgt = oc.window(soil_gt, soil.shape[1], soil.shape[0], bbox)
        want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
        for r in range(18):
            c, k = divmod(r, 9)
            p = tmp_path / ("cn_rasters_%s" % conds[c]) / ("cn_%s_%s_%d.tif" % (hcs[k // 3], arcs[k % 3], bid))
            assert np.array_equal(np.array(Image.open(str(p))), want[r]), p

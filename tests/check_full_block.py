#!/usr/bin/env python3
"""One-off end-to-end check on a GPU box (not collected by pytest): a full-size block of the real
VRT shape (36001 x 36001, noisy "natural" landcover in DEFLATE tiles, soil with and without dual
classes) through bin/gcn10; ALL 18 output GeoTIFFs are decoded by libtiff (Pillow) and compared
with the oracle on three bands of rows: the first 300, 600 in the middle, the last 300 (the
last tile row is 161 rows high).  Exit code 1 on any difference."""
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import cn_oracle_c as oc  # noqa: E402
from oracle import cn_oracle_np as onp  # noqa: E402
from tests import tiffutil  # noqa: E402

Image.MAX_IMAGE_PIXELS = None
LOOKUPS = os.path.join(ROOT, "tests", "golden", "lookups")
CONDS, HCS, ARCS = ("drained", "undrained"), ("p", "f", "g"), ("i", "ii", "iii")


def main():
    size, px = 36001, 8.3333333333330430e-05
    esa, _, coarse, _ = bench.synth_block(11, size, "natural")
    hs = coarse.shape[0]
    coarse[:, : hs // 2][coarse[:, : hs // 2] >= 11] -= 10          # western half: no dual classes
    egt = [0.0, px, 0.0, 3.0, 0.0, -px]
    sgt = [0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs]
    tabs = np.stack([oc.load_lookup_table(os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc)))[0]
                     for hc in onp.HCS for arc in onp.ARCS])
    with tempfile.TemporaryDirectory(dir="/tmp") as wd:
        tiffutil.write_tiff(os.path.join(wd, "esa.tif"), esa, gt=egt, compression=8, tile=(1024, 1024))
        tiffutil.write_tiff(os.path.join(wd, "soil.tif"), coarse, gt=sgt, compression=5, rows_per_strip=16)
        tiffutil.write_block_shapefile(os.path.join(wd, "blocks"), [(1, 0.0, 0.0, 3.0, 3.0)])
        with open(os.path.join(wd, "config.txt"), "w") as f:
            f.write("hysogs_data_path=%s/soil.tif\nesa_data_path=%s/esa.tif\nblocks_shp_path=%s/blocks.shp\n"
                    "lookup_table_path=%s\nlog_dir=%s/logs\n" % (wd, wd, wd, LOOKUPS, wd))
        p = subprocess.run([os.path.join(ROOT, "bin", "gcn10"), "-c", "config.txt"], cwd=wd, capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stderr[-2000:])
            sys.exit(1)
        xo, yo, W, H, gt = oc.window(egt, size, size, [0.0, 0.0, 3.0, 3.0])
        sxo, syo, hsx, hsy, sg = oc.window(sgt, hs, hs, [0.0, 0.0, 3.0, 3.0])
        bands = [(0, 300), (17800, 600), (H - 300, 300)]
        want = {}
        for y0, n in bands:
            want[y0] = oc.process_block_mem(esa[yo + y0:yo + y0 + n, xo:xo + W],
                                            [gt[0], gt[1], 0.0, gt[3] + y0 * gt[5], 0.0, gt[5]],
                                            coarse[syo:syo + hsy, sxo:sxo + hsx], sg, tabs)
        bad = 0
        for r in range(18):
            c, k = divmod(r, 9)
            path = os.path.join(wd, "cn_rasters_%s" % CONDS[c], "cn_%s_%s_1.tif" % (HCS[k // 3], ARCS[k % 3]))
            im = np.array(Image.open(path))
            ok = im.shape == (H, W) and all(np.array_equal(im[y0:y0 + n], want[y0][r]) for y0, n in bands)
            print("raster %2d %-40s %s" % (r, os.path.relpath(path, wd), "equal to the oracle on 1200 rows" if ok else "DIFFERENT"), flush=True)
            bad += not ok
        log = open(os.path.join(wd, "logs", "rank_0.log")).read()
        print([l for l in log.splitlines() if "timing" in l][-1][:200])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

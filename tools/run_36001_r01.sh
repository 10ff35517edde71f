#!/bin/bash
set -e
timeout -k 10 900 python tools/bench_pipeline.py --pattern patches --blocks 3 --modes files --esa-compression 8 --real-vrt-pixel --keep --workdir /tmp/gcn10_36001 > gpurun_out/pipeline_36001.json
cut -c1-700 gpurun_out/pipeline_36001.json
# spot check: decode one output raster fully with libtiff and compare a window with the oracle
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from PIL import Image
Image.MAX_IMAGE_PIXELS = None
from oracle import cn_oracle_c as oc
from tests.conftest import LOOKUPS
from oracle import cn_oracle_np as onp
wd = "/tmp/gcn10_36001"
os.system("cd %s && %s/bin/gcn10 -c config.txt -o > /dev/null 2>&1" % (wd, os.environ.get("GRAFT_REPO_ROOT", ".")))
esa = np.array(Image.open(wd + "/esa.tif"))
soil = np.array(Image.open(wd + "/soil.tif"))
print("esa", esa.shape, "soil", soil.shape)
tabs = np.stack([oc.load_lookup_table(os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc)))[0] for hc in onp.HCS for arc in onp.ARCS])
px = 8.3333333333330430e-05
egt = [0.0, px, 0.0, 3.0, 0.0, -px]
hs = soil.shape[0]
sgt = [0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs]
bbox = [3.0, 0.0, 6.0, 3.0]     # block 2
xo, yo, W, H, gt = oc.window(egt, esa.shape[1], esa.shape[0], bbox)
sxo, syo, hsx, hsy, sg = oc.window(sgt, soil.shape[1], soil.shape[0], bbox)
print("window", xo, yo, W, H)
rows = slice(17000, 17600)
want = oc.process_block_mem(esa[yo:yo + H, xo:xo + W][rows], [gt[0], gt[1], 0, gt[3] + 17000 * gt[5], 0, gt[5]], soil[syo:syo + hsy, sxo:sxo + hsx], sg, tabs)
ok = True
for r, (c, hc, arc) in ((0, ("drained", "p", "i")), (13, ("undrained", "f", "ii")), (17, ("undrained", "g", "iii"))):
    im = np.array(Image.open("%s/cn_rasters_%s/cn_%s_%s_2.tif" % (wd, c, hc, arc)))
    same = np.array_equal(im[rows], want[r])
    print("raster", r, im.shape, "rows 17000..17600 equal oracle:", same)
    ok &= same
sys.exit(0 if ok else 1)
PY

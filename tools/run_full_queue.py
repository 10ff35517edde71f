#!/usr/bin/env python3
"""BASELINE config 3 as worded: the FULL esa_extent_blocks queue through bin/gcn10 (GPU only).

All 2651 block IDs of the reference's own shapefile (tests/golden/blocks/esa_extent_blocks.shp, a
byte-identical copy of /root/reference/blocks/) in shapefile mode -- no -l list; the mode in which
the reference crashes (src/raster.c:97) -- against a coarse synthetic global
landcover / soil pair (0.02 degree landcover pixels: 18000 x 7200; soil 25x coarser), first with a
single lookup (--lookups g_ii --conditions drained), then with all 18 rasters.  Checks: every ID
3..2653 taken exactly once, the expected files exist, sampled rasters equal the oracle's.
Prints one JSON line.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from gcn10_amd import host  # noqa: E402
from oracle import cn_oracle_c as oc  # noqa: E402
from tests import tiffutil  # noqa: E402

ESA_CLASSES = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], dtype=np.uint8)
HSG_CODES = np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], dtype=np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--px-deg", type=float, default=0.02)
    ap.add_argument("--samples", type=int, default=12)
    ap.add_argument("--keep", default="")
    a = ap.parse_args()
    tmp = a.keep or tempfile.mkdtemp(prefix="gcn10_queue_")
    os.makedirs(tmp, exist_ok=True)
    shp = os.path.join(ROOT, "tests", "golden", "blocks", "esa_extent_blocks.shp")
    lookups = os.path.join(ROOT, "tests", "golden", "lookups")
    ids, bbox = host.read_blocks_shapefile(shp)
    bbox = np.array(bbox)
    W, H = int(round(360 / a.px_deg)), int(round(144 / a.px_deg))          # lon -180..180, lat 84..-60
    rng = np.random.default_rng(3)
    small = rng.choice(ESA_CLASSES, size=(H // 10, W // 10))
    esa = np.repeat(np.repeat(small, 10, axis=0), 10, axis=1)
    flip = rng.random(esa.shape) < 0.05
    esa = np.where(flip, rng.choice(ESA_CLASSES, size=esa.shape), esa).astype(np.uint8)
    soil = rng.choice(HSG_CODES, size=(H // 25, W // 25)).astype(np.uint8)
    esa_gt = [-180.0, a.px_deg, 0.0, 84.0, 0.0, -a.px_deg]
    soil_gt = [-180.0, a.px_deg * 25, 0.0, 84.0, 0.0, -a.px_deg * 25]
    tiffutil.write_tiff(os.path.join(tmp, "esa.tif"), esa, gt=esa_gt, compression=8, tile=(1024, 1024))
    tiffutil.write_tiff(os.path.join(tmp, "soil.tif"), soil, gt=soil_gt, compression=5, rows_per_strip=8)
    tables = host.load_all_lookup_tables(lookups)
    rec = {"blocks": len(ids), "landcover_px": [W, H], "runs": []}
    for name, flags, n_out, sel in (("single lookup (g_ii, drained)", ["--lookups", "g_ii", "--conditions", "drained"], 1, [7]),
                                    ("all 18 rasters", [], 18, list(range(18)))):
        work = os.path.join(tmp, "run%d" % n_out)
        os.makedirs(work, exist_ok=True)
        with open(os.path.join(work, "config.txt"), "w") as f:
            f.write("hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
                    % (os.path.join(tmp, "soil.tif"), os.path.join(tmp, "esa.tif"), shp, lookups,
                       os.path.join(work, "logs")))
        t0 = time.time()
        out = subprocess.run([os.path.join(ROOT, "bin", "gcn10"), "-c", "config.txt", "-o"] + flags, cwd=work,
                             capture_output=True, text=True, timeout=3000)
        wall = time.time() - t0
        logs = "".join(open(os.path.join(work, "logs", f)).read() for f in sorted(os.listdir(os.path.join(work, "logs"))))
        done = re.findall(r"completed condition for (\d+): ", logs)
        per_id = {}
        for d in done:
            per_id[int(d)] = per_id.get(int(d), 0) + 1
        n_files = sum(len(os.listdir(os.path.join(work, d))) for d in os.listdir(work) if d.startswith("cn_rasters_"))
        # sampled blocks against the oracle
        bad = 0
        pick = rng.choice(len(ids), size=a.samples, replace=False)
        for i in pick:
            xo, yo, w_, h_, gt = oc.window(esa_gt, W, H, bbox[i])
            sxo, syo, hsx, hsy, sgt = oc.window(soil_gt, soil.shape[1], soil.shape[0], bbox[i])
            want = oc.process_block_mem(esa[yo:yo + h_, xo:xo + w_], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
            for r in sel:
                cond = ("drained", "undrained")[r // 9]
                hc, arc = ("p", "f", "g")[(r % 9) // 3], ("i", "ii", "iii")[r % 3]
                p = os.path.join(work, "cn_rasters_%s" % cond, "cn_%s_%s_%d.tif" % (hc, arc, ids[i]))
                if not os.path.exists(p) or not np.array_equal(np.array(Image.open(p)), want[r]):
                    bad += 1
        timing = re.findall(r"timing: \d+ blocks.*", logs)
        host_cpu = re.findall(r"timing: host cpu seconds.*", logs)
        rec["runs"].append({
            "what": name, "rc": out.returncode, "wall_s": round(wall, 2),
            "log_says_processing": bool(re.search(r"processing %d blocks from shapefile" % len(ids), logs)),
            "log_says_processed": bool(re.search(r"processed %d blocks" % len(ids), logs)),
            "ids_completed": len(per_id), "ids_min": min(per_id) if per_id else None, "ids_max": max(per_id) if per_id else None,
            "every_id_exactly_n_rasters_once": all(v == n_out for v in per_id.values()) and sorted(per_id) == sorted(ids),
            "files": n_files, "files_expected": len(ids) * n_out,
            "sampled_blocks": int(a.samples), "sampled_rasters_differing_from_oracle": bad,
            "timing_line": timing[-1][:600] if timing else None,
            "host_cpu_line": host_cpu[-1][:300] if host_cpu else None,
            "stderr_tail": out.stderr[-300:] if out.returncode else ""})
    ok = all(r["rc"] == 0 and r["every_id_exactly_n_rasters_once"] and r["files"] == r["files_expected"]
             and r["sampled_rasters_differing_from_oracle"] == 0 and r["log_says_processed"] for r in rec["runs"])
    rec["ok"] = ok
    print(json.dumps(rec))
    if not a.keep:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

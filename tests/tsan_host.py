#!/usr/bin/env python3
"""`make tsan-host`: the threaded host program under ThreadSanitizer, against tests/stub_gpu (no GPU).

Builds bin-like executables from gcn10_amd/csrc/host/*.c with -fsanitize=thread, points GCN10_GPU_LIB
at the asynchronous host stub (tests/stub_gpu/stub_gpu.c, also instrumented), and drives 3 block
workers (each with its input thread, round 3) x 8 small blocks through gcn10_run in five input / sink modes.  Outputs are compared with the oracle
(the stub computes real rasters), and any ThreadSanitizer report fails the run.
TEST INFRASTRUCTURE ONLY: the stub is never a fallback of the product.
"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from gcn10_amd import host  # noqa: E402
from oracle import cn_oracle_c as oc  # noqa: E402
from tests import tiffutil  # noqa: E402

ESA_GT = [10.0, 0.001, 0.0, 50.0, 0.0, -0.001]
SOIL_GT = [9.9875, 0.025, 0.0, 50.0125, 0.0, -0.025]


def main():
    out_log = sys.argv[1] if len(sys.argv) > 1 else ""
    tmp = tempfile.mkdtemp(prefix="gcn10_tsan_")
    exe, stub = os.path.join(tmp, "gcn10_tsan"), os.path.join(tmp, "libstub_gpu.so")
    src = sorted(glob.glob(os.path.join(ROOT, "gcn10_amd", "csrc", "host", "*.c")))
    common = ["gcc", "-D_GNU_SOURCE", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread", "-fPIC",
              "-I" + os.path.join(ROOT, "include"), "-pthread"]
    subprocess.run(common + ["-std=c99", "-ffp-contract=off", "-o", exe] + src + ["-lm", "-lz", "-ldl"], check=True)
    subprocess.run(common + ["-std=c11", "-shared", "-o", stub, os.path.join(ROOT, "tests", "stub_gpu", "stub_gpu.c"), "-lz"],
                   check=True)

    rng = np.random.default_rng(11)
    H, W = 1600, 2400
    esa = np.repeat(np.repeat(rng.choice(np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], np.uint8),
                                         size=(H // 20, W // 20)), 20, axis=0), 20, axis=1)
    esa = np.where(rng.random(esa.shape) < 0.1, rng.integers(0, 256, esa.shape), esa).astype(np.uint8)
    soil = rng.choice(np.array([0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 255], np.uint8), size=(H // 25 + 2, W // 25 + 2))
    tiffutil.write_tiff(os.path.join(tmp, "esa.tif"), esa, gt=ESA_GT, compression=8, tile=(512, 512))
    # round 3: uncompressed tiles (staged raw, untiled on the "GPU") and DEFLATE tiles with TIFF predictor 2
    tiffutil.write_tiff(os.path.join(tmp, "esa_raw.tif"), esa, gt=ESA_GT, compression=1, tile=(256, 128))
    tiffutil.write_tiff(os.path.join(tmp, "esa_p2.tif"), esa, gt=ESA_GT, compression=8, predictor=2, tile=(512, 256))
    tiffutil.write_tiff(os.path.join(tmp, "soil.tif"), soil, gt=SOIL_GT, compression=5, rows_per_strip=8)
    blocks = []
    for i in range(8):          # 8 blocks of 0.6 x 0.8 degrees = 600 x 800 px, 4 across x 2 down
        x0, y0 = 10.0 + 0.6 * (i % 4), 50.0 - 0.8 * (i // 4)
        blocks.append((201 + i, x0, y0 - 0.8, x0 + 0.6, y0))
    tiffutil.write_block_shapefile(os.path.join(tmp, "blocks"), blocks)
    lookups = os.path.join(ROOT, "tests", "golden", "lookups")
    tables = host.load_all_lookup_tables(lookups)
    report = []
    failed = False
    for mode, extra in (("gpu_deflate=1 gpu_inflate=1", "gpu_deflate=1\ngpu_inflate=1\n"),
                        ("gpu_deflate=0 gpu_inflate=0", "gpu_deflate=0\ngpu_inflate=0\n"),
                        ("gpu_deflate=1 gpu_inflate=0 lookups=g_ii", "gpu_deflate=1\ngpu_inflate=0\nlookups=g_ii\nconditions=undrained\n"),
                        ("gpu_deflate=1 gpu_inflate=1 raw landcover tiles", "gpu_deflate=1\ngpu_inflate=1\nesa=esa_raw.tif\n"),
                        ("gpu_deflate=1 gpu_inflate=1 predictor-2 landcover, prefetch_blocks=0",
                         "gpu_deflate=1\ngpu_inflate=1\nprefetch_blocks=0\nesa=esa_p2.tif\n")):
        work = os.path.join(tmp, "run_" + str(len(report)))
        os.makedirs(work)
        esa_name = "esa.tif"
        for ln in extra.splitlines():
            if ln.startswith("esa="):
                esa_name = ln[4:]
        extra = "".join(ln + "\n" for ln in extra.splitlines() if not ln.startswith("esa="))
        with open(os.path.join(work, "config.txt"), "w") as f:
            f.write("hysogs_data_path=%s\nesa_data_path=%s\nblocks_shp_path=%s\nlookup_table_path=%s\nlog_dir=%s\n"
                    "strip_rows=256\nio_threads=4\n%s" % (os.path.join(tmp, "soil.tif"), os.path.join(tmp, esa_name),
                                                          os.path.join(tmp, "blocks.shp"), lookups,
                                                          os.path.join(work, "logs"), extra))
        env = dict(os.environ, GCN10_GPU_LIB=stub, GCN10_OVERSUBSCRIBE="1", GCN10_NO_NUMA_BIND="1",
                   TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66")
        out = subprocess.run([exe, "-c", "config.txt", "--gpus", "3"], cwd=work, capture_output=True, text=True, env=env,
                             timeout=1200)
        races = out.stderr.count("WARNING: ThreadSanitizer")
        bad = 0
        sel = [9 + 7] if "lookups" in mode else range(18)
        for bid, *bbox in blocks:
            xo, yo, w_, h_, gt = oc.window(ESA_GT, W, H, bbox)
            sxo, syo, hsx, hsy, sgt = oc.window(SOIL_GT, soil.shape[1], soil.shape[0], bbox)
            want = oc.process_block_mem(esa[yo:yo + h_, xo:xo + w_], gt, soil[syo:syo + hsy, sxo:sxo + hsx], sgt, tables)
            for r in sel:
                p = os.path.join(work, "cn_rasters_%s" % ("drained", "undrained")[r // 9],
                                 "cn_%s_%s_%d.tif" % (("p", "f", "g")[(r % 9) // 3], ("i", "ii", "iii")[r % 3], bid))
                if not os.path.exists(p) or not np.array_equal(np.array(Image.open(p)), want[r]):
                    bad += 1
        line = "tsan-host [%s]: exit %d, ThreadSanitizer reports %d, rasters differing from the oracle %d of %d, 3 workers x 8 blocks" % (
            mode, out.returncode, races, bad, len(blocks) * len(list(sel)))
        print(line)
        report.append(line)
        if out.returncode != 0 or races or bad:
            failed = True
            report.append(out.stderr[-6000:])
            print(out.stderr[-6000:])
    if out_log:
        with open(out_log, "w") as f:
            f.write("\n".join(report) + "\n")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end timing of bin/gcn10 on synthetic full-size blocks (BASELINE config 3/5 shape).

Builds, under --workdir, an uncompressed tiled landcover GeoTIFF of N blocks of SIZE^2 pixels
side by side, a 25x coarser LZW soil GeoTIFF and the block shapefile, then runs the program
(a) with GCN10_SINK=null: file decode + pinned staging + H2D + fused kernel + D2H, no encode,
(b) with the real sink: + 256x256 tile DEFLATE on the host pool + GeoTIFF files.
Prints one JSON line.  PCIe-inclusive numbers for DESIGN.md; never bench.py's `value`.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402
from tests import tiffutil  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=36000)
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--workdir", default="/tmp/gcn10_pipeline_bench")
    ap.add_argument("--strip-rows", type=int, default=0, help="0 = the program's choice")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--modes", default="null,files")
    ap.add_argument("--pattern", default="patches")
    ap.add_argument("--reuse", action="store_true", help="reuse the world already in --workdir")
    ap.add_argument("--keep", action="store_true", help="do not delete --workdir at the end")
    ap.add_argument("--deflate-level", type=int, default=0)
    ap.add_argument("--gpu-deflate", type=int, default=2)
    ap.add_argument("--workers-per-gpu", type=int, default=0)
    ap.add_argument("--gpu-inflate", type=int, default=1)
    ap.add_argument("--io-threads", type=int, default=0, help='config key "io_threads" (0 = the program\'s choice)')
    ap.add_argument("--lookups", default="", help='config key "lookups" (e.g. g_ii): BASELINE config 3, single lookup')
    ap.add_argument("--conditions", default="", help='config key "conditions" (drained | undrained | both)')
    ap.add_argument("--repeat", type=int, default=1,
                    help="list every block geometry this many times under different ids (long runs on a small world)")
    ap.add_argument("--dual-soil-fraction", type=float, default=-1.0,
                    help="fraction of the soil cells that keep a dual class (11..14); the others become 1..4. "
                         "Default: the bench distribution (40 %% dual), unlike most of the real world")
    ap.add_argument("--real-vrt-pixel", action="store_true",
                    help="use the shipped VRT's pixel size 8.3333333333330430e-05: 3-degree blocks become "
                         "36001 px wide (SURVEY.md section 7), rows are not 16-byte aligned")
    ap.add_argument("--esa-compression", type=int, default=1,
                    help="TIFF compression of the landcover input: 1 none, 8 DEFLATE (like the ESA COGs), 5 LZW")
    ap.add_argument("--esa-predictor", type=int, default=1, help="TIFF predictor of the landcover input (2 = horizontal differencing)")
    a = ap.parse_args()
    wd = a.workdir
    size, nb = a.size, a.blocks
    reuse = a.reuse and os.path.exists(os.path.join(wd, "esa.tif"))
    if not reuse:
        shutil.rmtree(wd, ignore_errors=True)
        os.makedirs(wd)
    px = 3.0 / size
    if a.real_vrt_pixel:
        size, px = 36001, 8.3333333333330430e-05
    t0 = time.time()
    if not reuse:
        build_world(a, wd, size, nb, px)
    else:
        # the block list is cheap and depends on --repeat: always as asked
        tiffutil.write_block_shapefile(os.path.join(wd, "blocks"),
                                       [(rep * nb + i + 1, 3.0 * i, 0.0, 3.0 * (i + 1), 3.0)
                                        for rep in range(a.repeat) for i in range(nb)])
    with open(os.path.join(wd, "config.txt"), "w") as f:
        f.write("hysogs_data_path=%s/soil.tif\nesa_data_path=%s/esa.tif\nblocks_shp_path=%s/blocks.shp\n"
                "lookup_table_path=%s\nlog_dir=%s/logs\nstrip_rows=%d\ndeflate_level=%d\ngpu_deflate=%d\n"
                "workers_per_gpu=%d\ngpu_inflate=%d\nio_threads=%d\n%s%s"
                % (wd, wd, wd, os.path.join(ROOT, "tests", "golden", "lookups"), wd, a.strip_rows,
                   a.deflate_level, a.gpu_deflate, a.workers_per_gpu, a.gpu_inflate, a.io_threads,
                   "lookups=%s\n" % a.lookups if a.lookups else "", "conditions=%s\n" % a.conditions if a.conditions else ""))
    build_s = time.time() - t0
    run_modes(a, wd, size, nb, build_s)


def build_world(a, wd, size, nb, px):
    # landcover: nb blocks side by side (lon 0..3*nb, lat 0..3), written tile-wise without
    # holding more than one block in memory
    esa1, _, coarse1, _ = bench.synth_block(1, size, a.pattern)
    esa = np.concatenate([esa1] * nb, axis=1) if nb > 1 else esa1
    del esa1
    tiffutil.write_tiff(os.path.join(wd, "esa.tif"), esa, gt=[0.0, px, 0.0, 3.0, 0.0, -px],
                        compression=a.esa_compression, predictor=a.esa_predictor, tile=(1024, 1024), bigtiff=esa.size > 3 * 2**30)
    del esa
    hs = coarse1.shape[0]
    if a.dual_soil_fraction >= 0.0:
        # dual classes only inside a few large patches, as wet lowlands are
        rng = np.random.default_rng(12)
        keep = np.repeat(np.repeat(rng.random(((hs + 95) // 96, (hs + 95) // 96)) < a.dual_soil_fraction, 96, axis=0),
                         96, axis=1)[:hs, :hs]
        dual = coarse1 >= 11
        coarse1 = np.where(dual & ~keep, coarse1 - 10, coarse1).astype(np.uint8)
    soil = np.concatenate([coarse1] * nb, axis=1) if nb > 1 else coarse1
    tiffutil.write_tiff(os.path.join(wd, "soil.tif"), soil, gt=[0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs],
                        compression=8, rows_per_strip=64)
    tiffutil.write_block_shapefile(os.path.join(wd, "blocks"),
                                   [(rep * nb + i + 1, 3.0 * i, 0.0, 3.0 * (i + 1), 3.0)
                                    for rep in range(a.repeat) for i in range(nb)])


def run_modes(a, wd, size, nb, build_s):
    n_rasters = 18
    if a.lookups or a.conditions:
        n_rasters = (len([x for x in a.lookups.split(",") if x]) if a.lookups and a.lookups != "all" else 9) * \
            (1 if a.conditions in ("drained", "undrained") else 2)
    res = {"size": size, "blocks": nb, "rasters_per_block": n_rasters, "strip_rows": a.strip_rows, "gpus": a.gpus, "pattern": a.pattern, "gpu_deflate": a.gpu_deflate, "gpu_inflate": a.gpu_inflate, "workers_per_gpu": a.workers_per_gpu, "esa_compression": a.esa_compression, "dual_soil_fraction": a.dual_soil_fraction,
           "world_build_seconds": round(build_s, 1), "modes": {}}
    for mode in a.modes.split(","):
        env = dict(os.environ)
        if mode == "null":
            env["GCN10_SINK"] = "null"
        shutil.rmtree(os.path.join(wd, "logs"), ignore_errors=True)
        t0 = time.time()
        # GCN10_BIN: another build of the program (same-box A/B against an earlier round: tools/r03/run_vs_r02.sh)
        out = subprocess.run([os.environ.get("GCN10_BIN") or os.path.join(ROOT, "bin", "gcn10"), "-c", "config.txt", "-o", "--gpus", str(a.gpus)],
                             cwd=wd, env=env, capture_output=True, text=True)
        wall = time.time() - t0
        log = open(os.path.join(wd, "logs", "rank_0.log")).read() if os.path.exists(os.path.join(wd, "logs", "rank_0.log")) else ""
        m = re.search(r"timing: (\d+) blocks, ([0-9.]+) s wall(?: \(([0-9.]+) s after start-up\))?", log)
        ms = re.search(r"timing: steady state ([0-9.]+) s per block", log)
        done = int(m.group(1)) if m else 0
        secs = float(m.group(2)) if m else wall
        steady = float(m.group(3)) if m and m.group(3) else None
        nbytes = 0
        for d in ("cn_rasters_drained", "cn_rasters_undrained"):
            p = os.path.join(wd, d)
            if os.path.isdir(p):
                nbytes += sum(os.path.getsize(os.path.join(p, f)) for f in os.listdir(p))
        mt = re.search(r"worker seconds: (.*)", log)
        mc = re.search(r"host cpu seconds: user ([0-9.]+), system ([0-9.]+), over [0-9.]+ s wall \(([0-9.]+) per block\).*?pinned host memory allocated ([0-9.]+) MB; peak resident set ([0-9.]+) MB", log)
        res["modes"][mode] = {"rc": out.returncode, "worker_seconds": mt.group(1) if mt else None, "blocks_done": done, "seconds": round(secs, 3),
                              "cn_gpx_per_s": round(done * size * size * n_rasters / secs / 1e9, 3) if secs else None,
                              "seconds_per_block": round(secs / done, 3) if done else None,
                              "seconds_after_startup": steady,
                              "steady_seconds_per_block": round(steady / done, 4) if done and steady else None,
                              # every worker's blocks after its first (which pays for allocations and pinning)
                              "after_first_block_seconds_per_block": float(ms.group(1)) if ms else None,
                              "steady_cn_gpx_per_s": round(done * size * size * n_rasters / steady / 1e9, 1) if steady else None,
                              "host_cpu_seconds_per_block": float(mc.group(3)) if mc else None,
                              "host_cpu_user_system": [float(mc.group(1)), float(mc.group(2))] if mc else None,
                              "pinned_MB": float(mc.group(4)) if mc else None, "peak_rss_MB": float(mc.group(5)) if mc else None,
                              "output_bytes": nbytes, "stderr_tail": out.stderr[-300:] if out.returncode else ""}
    print(json.dumps(res))
    for d in ("cn_rasters_drained", "cn_rasters_undrained", "logs"):
        shutil.rmtree(os.path.join(wd, d), ignore_errors=True)
    if not a.keep:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the curve-number hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic 36000 x 36000 uint8
block that is already resident in HBM: the x-expansion of the coarse soil
window (gcn10_gpu_prepare_tile) + the fused strip kernel over the whole block
(gcn10_gpu_cn_strip), through the C ABI of include/gcn10_gpu.h.

Workloads (BASELINE.json `configs`):
  config2                  one lookup (g_ii, "ARC-II"), drained: 1 CN raster, soil resample fused
                           (2.0016 B/px algorithmic)  -- the default, configs[1]
  config2-preresampled     the same raster from a full-resolution soil tile
                           (gcn10_gpu_calculate_cn, 3 B/px)
  config4                  all 9 lookups x {drained, undrained} fused: 18 rasters (19.0016 B/px)
  config4-drained          9 rasters of one drainage condition (10.0016 B/px)

value = CN Gpixels/s = block pixels x CN rasters produced x N / elapsed (max over ranks);
at N > 1 every rank owns its own block (weak scaling, no data-path collective).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level table)
K_G_II = 7                  # table index of default_lookup_g_ii.csv (hc=g -> 2, arc=ii -> 1)

WORKLOADS = {
    #  name                   (cond_mask, table_mask, preresampled)
    "config2": (1, 1 << K_G_II, False),
    "config2-preresampled": (1, 1 << K_G_II, True),
    "config4": (3, 0x1FF, False),
    "config4-drained": (1, 0x1FF, False),
}
ESA_CLASSES = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], dtype=np.uint8)
HSG_CODES = np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], dtype=np.uint8)


def synth_block(seed: int, size: int, pattern: str):
    """landcover from the 12-value class set (30% NoData 0,
    rest spread evenly), coarse soil window 1440^2 (ratio 25) from the 10-value code set."""
    rng = np.random.default_rng(seed)
    # byte -> class table: 77/256 = 30% NoData 0, the other 11 classes 16-17/256 each
    lut = np.zeros(256, dtype=np.uint8)
    lut[77:] = ESA_CLASSES[1:][(np.arange(179) * 11) // 179]
    if pattern == "patches":
        n = (size + 63) // 64
        small = lut[rng.integers(0, 256, size=(n, n), dtype=np.uint8)]
        esa = np.ascontiguousarray(np.repeat(np.repeat(small, 64, axis=0), 64, axis=1)[:size, :size])
    elif pattern == "natural":
        # between the extremes (tools/ only, not a bench.py workload): half the 256-px cells are
        # 64-px patches, half 8-px blobs, 3 % of all pixels flipped; one 4096^2 piece repeated
        p = min(size, 4096)
        n = (p + 63) // 64
        coarse_p = np.repeat(np.repeat(lut[rng.integers(0, 256, size=(n, n), dtype=np.uint8)], 64, axis=0), 64, axis=1)[:p, :p]
        m = (p + 7) // 8
        fine_p = np.repeat(np.repeat(lut[rng.integers(0, 256, size=(m, m), dtype=np.uint8)], 8, axis=0), 8, axis=1)[:p, :p]
        k = (p + 255) // 256
        pick = np.repeat(np.repeat(rng.random((k, k)) < 0.5, 256, axis=0), 256, axis=1)[:p, :p]
        piece = np.where(pick, coarse_p, fine_p)
        flip = rng.random((p, p)) < 0.03
        piece = np.where(flip, lut[rng.integers(0, 256, size=(p, p), dtype=np.uint8)], piece).astype(np.uint8)
        reps = (size + p - 1) // p
        esa = np.ascontiguousarray(np.tile(piece, (reps, reps))[:size, :size])
    else:
        # i.i.d. pixels; one random slab repeated down the block (the content repeats at
        # different addresses, so no cache can profit) keeps host-side generation short
        slab = lut[rng.integers(0, 256, size=(min(size, 2250), size), dtype=np.uint8)]
        esa = np.ascontiguousarray(np.tile(slab, ((size + slab.shape[0] - 1) // slab.shape[0], 1))[:size])
    hs = max(1, size // 25)
    coarse = rng.choice(HSG_CODES, size=(hs, hs)).astype(np.uint8)
    gt = [0.0, 3.0 / size, 0.0, 3.0, 0.0, -3.0 / size]
    soil_gt = [0.0, 3.0 / hs, 0.0, 3.0, 0.0, -3.0 / hs]
    return esa, gt, coarse, soil_gt


def cpu_baseline(workload: str, size: int):
    """Reference-shaped oracle on this box's host cores; separate process tree, before
    this process touches the GPU."""
    cond_mask, table_mask, _ = WORKLOADS[workload]
    n_out = bin(cond_mask).count("1") * bin(table_mask).count("1")
    # about 10-30 s of CPU work per worker: 16000 rows for one raster, 2000 rows for all 18
    rows = max(16, min(size, (16000 if n_out == 1 else 36000 // n_out)))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--width", str(size),
           "--rows", str(rows), "--cond-mask", str(cond_mask), "--table-mask", str(table_mask)]
    try:
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=900).stdout
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as exc:  # the baseline is a reported extra, never a reason to lose the bench line
        return {"value": None, "unit": "CN Gpx/s", "cores": 0, "kind": "port", "sample": "failed: %s" % exc}
    return {
        "value": round(r["multi"]["gpx_per_s"], 4), "unit": "CN Gpx/s", "cores": r["multi"]["procs"],
        "kind": "port",
        "sample": "%d independent worker processes (as mpirun -n %d), each %d rows x %d px of the "
                  "synthetic block, %d CN raster(s), oracle_process_block_subset (src/cn.c:218-290 "
                  "loop structure, per-raster malloc/memcpy/memset, I/O excluded), gcc -O3 no -march; "
                  "slowest worker %.1f s" % (r["multi"]["procs"], r["multi"]["procs"], rows, size, n_out,
                                             r["multi"]["worker_seconds_max"]),
        "single_core_value": round(r["single"]["gpx_per_s"], 4),
        "host_cores_available": r["cores_available"],
        "host_cpu": _host_cpu(),
    }


def _host_cpu():
    """Model name and socket count of the box's host CPU (SURVEY.md 8d asks for them beside the baseline)."""
    model, sockets = "", set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    sockets.add(line.split(":", 1)[1].strip())
    except OSError:
        pass
    return {"model": model, "sockets": len(sockets) or None}


def traffic_from_profiles(workload: str):
    """HBM bytes per launch from the committed PMC pass (profiles/pmc_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(workload)
        return rec["hbm_bytes_per_launch"] if rec else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=36000, help="block edge in pixels (36000 = BASELINE)")
    ap.add_argument("--pattern", default="iid", choices=["iid", "patches"])
    ap.add_argument("--strip-rows", type=int, default=0, help="0 = whole block in one launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short extra config4 measurement")
    args = ap.parse_args()

    from gcn10_amd import shard
    rank, local_rank, world = shard.world_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    cond_mask, table_mask, preresampled = WORKLOADS[args.workload]
    n_out = bin(cond_mask).count("1") * bin(table_mask).count("1")
    size = args.size

    # CPU baseline first: rank 0, N = 1 only, in a child process tree that never sees the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, size)

    from gcn10_amd import gpu, host
    grp = shard.Group()                     # nccl (= RCCL) when WORLD_SIZE > 1, nothing otherwise
    n_dev = gpu.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if local_rank >= n_dev and os.environ.get("GCN10_DIST_BACKEND") != "gloo":
        raise SystemExit("rank %d has no GPU (only %d visible)" % (local_rank, n_dev))
    eng = gpu.Engine(local_rank % n_dev)      # modulo only matters for the gloo rehearsal on one GPU
    info = eng.device_info()
    tables = host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups"))
    eng.set_tables(tables)

    esa, gt, coarse, soil_gt = synth_block(1 + rank, size, args.pattern)
    hs = coarse.shape[0]
    ci, cj = host.build_index_maps(gt, soil_gt, size, size, hs, hs)
    npix = size * size
    d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
    del esa
    outs = [None] * 18
    out_bufs = []
    for r in range(18):
        if (cond_mask >> (r // 9)) & 1 and (table_mask >> (r % 9)) & 1:
            b = eng.alloc(npix)
            out_bufs.append(b)
            outs[r] = b.ptr
    d_fine = None
    if preresampled:
        d_fine = eng.alloc(npix)
        eng.resample(d_coarse.ptr, hs, hs, d_ci.ptr, d_cj.ptr, size, size, d_fine.ptr)
        eng.sync()

    strip = args.strip_rows or size
    ev = [(eng.event_create(), eng.event_create()) for _ in range(args.steps)]

    def step(i_timed=None):
        if preresampled:
            if i_timed is not None:
                eng.event_record(ev[i_timed][0])
            eng.calculate_cn(d_esa.ptr, d_fine.ptr, npix, K_G_II, out_bufs[0].ptr)
            if i_timed is not None:
                eng.event_record(ev[i_timed][1])
            return
        eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
        whole = strip >= size
        if i_timed is not None and not whole:
            eng.event_record(ev[i_timed][0])
        for y0 in range(0, size, strip):
            rows = min(strip, size - y0)
            ptrs = [p + y0 * size if p else None for p in outs]
            if i_timed is not None and whole:
                # events carried by the dispatch itself: the kernel's own duration
                eng.time_next_strip(ev[i_timed][0], ev[i_timed][1])
            eng.cn_strip(d_esa.at(y0 * size), size, rows, d_cj.at(4 * y0), cond_mask, table_mask, ptrs)
        if i_timed is not None and not whole:
            eng.event_record(ev[i_timed][1])

    def full_sync():
        eng.device_sync()
        if "torch" in sys.modules:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    full_sync()
    grp.barrier()
    full_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    full_sync()
    grp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = grp.max(elapsed)

    kernel_ms = [eng.elapsed_ms(a, b) for a, b in ev]
    kname = "calculate_cn_kernel<true>" if preresampled else eng.last_kernel_name()
    launches = 1 if preresampled else (size + strip - 1) // strip
    avg_launch_s = float(np.mean(kernel_ms)) / 1e3 / launches
    if preresampled:
        alg_bytes = 3 * npix
    else:
        alg_bytes = gpu.strip_algorithmic_bytes(size, size, hs, hs, cond_mask, table_mask) / launches
    achieved = alg_bytes / avg_launch_s / 1e9

    also = None
    if world == 1 and not args.no_also and args.workload == "config2" and not preresampled:
        # the product's per-block pass (18 rasters fused), measured outside the timed region
        try:
            extra = [eng.alloc(npix) for _ in range(17)]
            all_outs = [out_bufs[0].ptr] + [b.ptr for b in extra]
            e0, e1 = eng.event_create(), eng.event_create()
            n_rep = 5
            eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, all_outs)
            eng.sync()
            eng.event_record(e0)
            for _ in range(n_rep):
                eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, all_outs)
            eng.event_record(e1)
            eng.sync()
            ms = eng.elapsed_ms(e0, e1) / n_rep
            b18 = gpu.strip_algorithmic_bytes(size, size, hs, hs, 3, 0x1FF)
            also = {"workload": "config4 (18 rasters fused, kernel only)", "kernel": eng.last_kernel_name(),
                    "ms_per_launch": round(ms, 4), "cn_gpx_per_s": round(npix * 18 / ms / 1e6, 2),
                    "achieved_GBps": round(b18 / ms / 1e6, 1), "frac_of_peak": round(b18 / ms / 1e6 / HBM_PEAK_GBS, 4)}
            for b in extra:
                b.close()
        except gpu.Gcn10GpuError as exc:
            also = {"error": str(exc)}

    if rank == 0:
        value = npix * n_out * world / elapsed * args.steps / 1e9
        line = {
            "metric": "CN Gpixels/sec", "value": round(value, 3), "unit": "Gpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: one %dx%d uint8 landcover block per GPU, %s, %d CN raster(s) per step; "
                                   "soil %s" % (args.workload, size, size,
                                                "lookup g_ii (ARC-II)" if n_out < 9 else "all 9 lookups",
                                                n_out,
                                                "pre-resampled full-res tile" if preresampled
                                                else "window %dx%d resampled in-kernel" % (hs, hs)),
                       "pattern": args.pattern, "strip_rows": strip, "blocks_per_step_per_gpu": 1,
                       "device": info["name"], "cus": info["cus"]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the committed PMC passes were taken at the default shape only
                         "traffic": traffic_from_profiles(args.workload)
                         if (size == 36000 and strip == size and args.pattern == "iid") else None,
                         "kernel": kname, "algorithmic_bytes_per_launch": int(alg_bytes),
                         "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                         "median_launch_ms": round(float(np.median(kernel_ms)) / launches, 4),
                         "min_launch_ms": round(float(np.min(kernel_ms)) / launches, 4)},
            "cpu_baseline": cpu,
        }
        if also:
            line["also"] = also
        print(json.dumps(line))
    grp.close()
    eng.close()


if __name__ == "__main__":
    main()

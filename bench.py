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

value = CN Gpixels/s = block pixels x CN rasters produced x N / whole-job time, where the
whole-job time runs from the first rank's start to the last rank's end (CLOCK_MONOTONIC is
common to the processes of one node).  At N > 1 every rank owns its own block on its own GPU
(weak scaling): the path has no collective (src/main.c:171 -- independent ranks, one barrier),
so ranks meet only at a barrier before and after the timed steps, through files
(gcn10_amd/shard.py) -- no RCCL, no torch.  `python bench.py --gpus N` without a launcher starts
the N rank processes itself, before anything touches a GPU.

Before the timed region (all of it untimed set-up, recorded in the JSON line): the placement calibration
of the one-raster workload (roofline.placement: 24 candidate rasters held together, positions and launch
shapes inside each, three placements of the landcover; --no-tune skips it), --pre-warm-ms of the same
step (config.pre_warm_steps_untimed: the card's launch times settle only after some tens of
milliseconds of load), then the W warm-up steps, a barrier, and the K timed steps.
Environment: GCN10_BENCH_AB=1 adds `ab_prepare_tile` (per-step kernel times of the timed region; the same
launch with and without gcn10_gpu_prepare_tile in front, measured afterwards); GCN10_BENCH_ENGINE names a
stand-in engine class for the launcher tests; GCN10_BENCH_OVERSUBSCRIBE=1 = --oversubscribe.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
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
    this process touches the GPU.  SURVEY.md 8(d): P = 1 and P = all physical cores (plus the
    one-GPU box's 16-core share), and the fused -march=native pass as a "best CPU" line."""
    cond_mask, table_mask, _ = WORKLOADS[workload]
    n_out = bin(cond_mask).count("1") * bin(table_mask).count("1")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--width", str(size),
           "--cond-mask", str(cond_mask), "--table-mask", str(table_mask)]
    try:
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=900).stdout
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as exc:  # the baseline is a reported extra, never a reason to lose the bench line
        return {"value": None, "unit": "CN Gpx/s", "cores": 0, "kind": "port", "sample": "failed: %s" % exc}
    head = r["headline"]
    return {
        "value": round(head["gpx_per_s"], 4), "unit": "CN Gpx/s", "cores": head["procs"],
        "kind": "port",
        "sample": "%d independent worker processes (as mpirun -n %d, src/main.c:171), each %d rows x %d px of "
                  "the synthetic block, %d CN raster(s), oracle_process_block_subset (src/cn.c:218-290 "
                  "loop structure, per-raster malloc/memcpy/memset, I/O excluded), gcc -O3 no -march; "
                  "slowest worker %.1f s; %s" % (head["procs"], head["procs"], head["rows_per_proc"], size, n_out,
                                                 head["worker_seconds_max"],
                                                 ("the box's cgroup CPU quota is %s CPUs: runs with more workers than "
                                                  "that (see `runs`) share the same quota and are not faster"
                                                  % r.get("cpu_quota_cores")) if r.get("cpu_quota_cores")
                                                 else "no cgroup CPU quota found"),
        "runs": r["runs"],                              # P = 1, the 16-core share, all physical cores
        "best_cpu": r.get("best_cpu"),                  # fused single pass, -O3 -march=native, all physical cores
        "single_core_value": round(r["runs"][0]["gpx_per_s"], 4),
        "host_cores_available": r["cores_available"], "host_physical_cores": r["physical_cores"],
        "cpu_quota_cores": r.get("cpu_quota_cores"),
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


def traffic_from_profiles(workload: str, field: str = "hbm_bytes_per_launch", kernel: str = ""):
    """HBM bytes per launch from the committed PMC pass (profiles/pmc_traffic.json), or None.
    field="kernel": the kernel instantiation that pass measured (it is the launch shape the calibration
    chose on the profiling box, not necessarily the one this run times)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
        rec = doc.get(workload)
        if kernel and workload == "config2":
            # round 3: the passes keep every single-raster variant the calibration launched
            by = doc.get("config2_by_kernel", {}).get(kernel)
            if by and by.get("hbm_bytes_per_launch"):
                return by["hbm_bytes_per_launch"] if field == "hbm_bytes_per_launch" else kernel
        return rec.get(field) if rec else None
    except Exception:
        return None


def _numa_node_of(pci_bus_id: str):
    try:
        with open("/sys/bus/pci/devices/%s/numa_node" % pci_bus_id.lower()) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def _bind_to_numa_node(node):
    """Run this rank on the cores next to its GPU (what bin/gcn10 does for its workers, pipeline.c)."""
    if node is None or node < 0:
        return None
    try:
        cpus = set()
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return len(cpus)
    except (OSError, ValueError):
        pass
    return None


def _stats(ms):
    return {"avg_ms": round(float(np.mean(ms)), 4), "median_ms": round(float(np.median(ms)), 4),
            "min_ms": round(float(np.min(ms)), 4)}


def _load_engine_class():
    """gcn10_amd.gpu.Engine -- or, for the launcher / aggregation tests only, the class named by
    GCN10_BENCH_ENGINE (module:Class under tests/); the bench line then says so in `data`."""
    spec = os.environ.get("GCN10_BENCH_ENGINE")
    if not spec:
        from gcn10_amd import gpu
        return gpu.Engine, gpu.device_count, False
    mod, _, cls = spec.partition(":")
    klass = getattr(importlib.import_module(mod), cls)
    return klass, klass.device_count, True


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes -- before this process
    imports the GPU library or touches a device -- and pass rank 0's JSON line on.  A rank that fails
    ends the job: the others are stopped instead of waiting for it at the barrier."""
    rdv = tempfile.mkdtemp(prefix="gcn10_rdv_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    out0 = open(os.path.join(rdv, "rank0.stdout"), "w+")
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       GCN10_RDV_DIR=rdv, MASTER_ADDR="127.0.0.1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        failed = []
        while True:
            rcs = [p.poll() for p in procs]
            failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if failed or all(rc == 0 for rc in rcs):
                break
            time.sleep(0.05)
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        out0.close()
        shutil.rmtree(rdv, ignore_errors=True)
    if failed:
        raise SystemExit("bench.py: rank(s) failed: %s" % ", ".join("rank %d rc %d" % b for b in failed))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=36000, help="block edge in pixels (36000 = BASELINE)")
    ap.add_argument("--pattern", default="iid", choices=["iid", "patches"])
    ap.add_argument("--strip-rows", type=int, default=0, help="0 = whole block in one launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra config4 and copy measurements")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one block per GPU (default); strong = ONE block, every GPU takes a band of "
                         "its rows (rows are independent given cj[], src/cn.c:218-232)")
    ap.add_argument("--no-tune", action="store_true",
                    help="skip the placement / launch-shape calibration of the one-raster workload")
    ap.add_argument("--tune-arenas", type=int, default=24,
                    help="candidate allocations the calibration chooses the raster buffer from")
    ap.add_argument("--tune-sources", type=int, default=3,
                    help="placements of the landcover block the calibration chooses from")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the comparison of the timed raster with a second variant's (untimed, after the measurement)")
    ap.add_argument("--pre-warm-ms", type=float, default=150.0,
                    help="untimed steps run for this long before the W warm-up steps (clock ramp after the set-up)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal: let ranks share GPUs (rank r uses device r mod visible devices)")
    args = ap.parse_args(argv)

    from gcn10_amd import shard
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args, argv)
        return
    rank, local_rank, world = shard.world_from_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d (run `python bench.py --gpus N` by itself, or under "
                         "torch.distributed.run with --nproc-per-node N)" % (world, args.gpus))

    cond_mask, table_mask, preresampled = WORKLOADS[args.workload]
    n_out = bin(cond_mask).count("1") * bin(table_mask).count("1")
    size = args.size

    # CPU baseline first: rank 0, N = 1 only, in a child process tree that never sees the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, size)

    from gcn10_amd import host
    Engine, device_count, fake_engine = _load_engine_class()
    from gcn10_amd import gpu       # strip_algorithmic_bytes: a host-side formula of the C ABI
    grp = shard.Group()             # files in a per-job directory; nothing at world size 1
    n_dev = device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    oversub = args.oversubscribe or os.environ.get("GCN10_BENCH_OVERSUBSCRIBE") == "1"
    if local_rank >= n_dev and not oversub:
        raise SystemExit("rank %d has no GPU (only %d visible); --oversubscribe shares GPUs for a rehearsal"
                         % (local_rank, n_dev))
    device = local_rank % n_dev
    eng = Engine(device)
    info = eng.device_info()
    bus = eng.pci_bus_id() if hasattr(eng, "pci_bus_id") else ""
    numa = _numa_node_of(bus) if bus else None
    bound = _bind_to_numa_node(numa) if world > 1 else None
    tables = host.load_all_lookup_tables(os.path.join(ROOT, "tests", "golden", "lookups"))
    eng.set_tables(tables)

    strong = args.scaling == "strong" and world > 1
    esa, gt, coarse, soil_gt = synth_block(1 if strong else 1 + rank, size, args.pattern)
    hs = coarse.shape[0]
    ci, cj = host.build_index_maps(gt, soil_gt, size, size, hs, hs)
    # strong scaling: this rank's band of rows of the one block (whole 16-row groups, so that every band
    # starts 16-byte aligned for any width); the soil window and index maps are shared by construction
    band = (0, size)
    if strong:
        edges = [min(size, (size * r // world + 15) // 16 * 16) for r in range(world)] + [size]
        band = (edges[rank], edges[rank + 1])
        esa = np.ascontiguousarray(esa[band[0]:band[1]])
        cj = np.ascontiguousarray(cj[band[0]:band[1]])
    rows_mine = band[1] - band[0]
    npix = size * rows_mine
    d_esa, d_coarse, d_ci, d_cj = eng.upload(esa), eng.upload(coarse), eng.upload(ci), eng.upload(cj)
    esa_host = esa          # kept until the calibration has had its chance to place it elsewhere
    del esa
    want_also = world == 1 and not args.no_also and args.workload == "config2" and not preresampled
    outs = [None] * 18
    out_bufs = []
    placement = None
    spare_rasters = []      # candidates of the calibration kept for the config-4 leg
    # (A/B switch: allocating the 17 extra rasters of the config4 leg BEFORE the calibration's candidates put
    # every candidate in a slow place in both processes that tried it -- 0.484 ms against 0.438-0.465 ms,
    # profiles/r02/bench_alloc_order_ab.txt -- so they are allocated after it)
    extra_first = os.environ.get("GCN10_BENCH_EXTRA_FIRST", "0") == "1"
    extra = []
    if extra_first and world == 1 and not args.no_also and args.workload == "config2" and not preresampled:
        extra = [eng.alloc(size * size) for _ in range(17)]
    tune = (n_out == 1 and not preresampled and not args.no_tune and not args.strip_rows
            and hasattr(eng, "tune_single_raster"))
    for r in range(18):
        if (cond_mask >> (r // 9)) & 1 and (table_mask >> (r % 9)) & 1:
            if tune:
                cands, spacers = [], []
                try:
                    # a one-raster strip is a 1R:1W stream whose rate depends on where the raster lies
                    # relative to the landcover -- within an allocation periodically in the distance (128 MiB),
                    # and from one allocation to the next by a few percent.  Let the
                    # library time the positions of one period inside each of a few candidate allocations
                    # (and its launch shapes); keep the best allocation, free the others.
                    slack = 160 << 20
                    eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
                    # (Rasters assembled from far-apart physical chunks through HIP's virtual memory management
                    # ran this kernel at 0.416-0.427 ms on boxes where every plain candidate runs at 0.48 ms, but
                    # nontemporal stores into such ranges were not reliably visible when the kernel had completed:
                    # profiles/r02/spread_allocator_hazard.txt.  Plain allocations only.)
                    # VRAM comes in two classes of region that alternate every ~16 GiB of allocation order, and
                    # a raster in the class the landcover is not in is written 6-12 % faster
                    # (tools/region_lab.hip): the candidates are all held until the choice is made, with a
                    # 2 GiB spacer after every fourth, so that twenty-four of them span ~47 GiB.
                    cands, spacers = [], []
                    for k in range(max(1, args.tune_arenas)):
                        try:
                            cands.append(eng.alloc(npix + slack))
                            if k % 4 == 3:
                                spacers.append(eng.alloc(2 << 30))
                        except Exception as exc:
                            sys.stderr.write("bench.py: candidate skipped (%s)\n" % exc)
                    if not cands:
                        raise RuntimeError("no candidate raster could be allocated")
                    tried = []
                    for c in cands:
                        _, ms, _ = eng.tune_single_raster(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask,
                                                          c.ptr, npix + slack, 16 << 20)
                        tried.append(round(ms, 4))
                    keep = int(np.argmin(tried))
                    # the seventeen next-best candidates become the other rasters of the config-4 leg (they
                    # are written, never read: what the calibration timed is what that kernel does to them)
                    order = [i for i in np.argsort(tried) if i != keep]
                    keep_extra = set(order[:17]) if want_also else set()
                    for i, c in enumerate(cands):
                        if i != keep and i not in keep_extra:
                            c.close()
                    spare_rasters = [cands[i] for i in order[:17] if i in keep_extra]
                    for sp in spacers:
                        sp.close()
                    b = cands[keep]
                    out_bufs.append(b)
                    # once more on the winner: leaves ITS best launch shape set in the context
                    best, best_ms, placement = eng.tune_single_raster(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask,
                                                                      table_mask, b.ptr, npix + slack, 16 << 20)
                    placement["allocations_tried_best_ms"] = tried
                    outs[r] = best
                    # ... and where the landcover lies matters too (some allocations read or write a few
                    # percent faster than others whatever their partner, profiles/r02/placement_probe2_*):
                    # the same block uploaded to one or two more places, each timed against the chosen raster
                    src_ms = [round(best_ms, 4)]
                    for k in range(max(0, args.tune_sources - 1)):
                        alt = eng.upload(esa_host)
                        a_best, a_ms, a_rep = eng.tune_single_raster(alt.ptr, size, rows_mine, d_cj.ptr, cond_mask,
                                                                     table_mask, b.ptr, npix + slack, 16 << 20)
                        src_ms.append(round(a_ms, 4))
                        if a_ms < best_ms * 0.995:
                            d_esa.close()
                            d_esa, best_ms, outs[r] = alt, a_ms, a_best
                        else:
                            alt.close()
                    # the shape left in the context belongs to the last call: set the winner's again
                    best, best_ms, rep = eng.tune_single_raster(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask,
                                                                table_mask, b.ptr, npix + slack, 16 << 20)
                    outs[r] = best
                    rep["allocations_tried_best_ms"] = tried
                    rep["landcover_placements_tried_best_ms"] = src_ms
                    placement = rep
                except Exception as exc:       # the calibration is an optimisation: never lose the bench line over it
                    sys.stderr.write("bench.py: placement calibration skipped (%s)\n" % exc)
                    for c in cands + spacers:
                        try:
                            c.close()
                        except Exception:
                            pass
                    out_bufs[:] = [x for x in out_bufs if x not in cands]
                    spare_rasters = []
                    eng.set_option("defaults", 0)
                    placement = {"error": str(exc)}
                    b = eng.alloc(npix)
                    out_bufs.append(b)
                    outs[r] = b.ptr
            else:
                b = eng.alloc(npix)
                out_bufs.append(b)
                outs[r] = b.ptr
    out0 = next(p for p in outs if p)
    del esa_host
    if want_also and not extra:
        # the extra config4 measurement writes 18 rasters: allocate the other 17 now, long before they are timed
        extra = list(spare_rasters[:17])
        extra += [eng.alloc(npix) for _ in range(17 - len(extra))]
    d_fine = None
    if preresampled:
        d_fine = eng.alloc(npix)
        eng.resample(d_coarse.ptr, hs, hs, d_ci.ptr, d_cj.ptr, size, rows_mine, d_fine.ptr)
        eng.sync()

    strip = args.strip_rows or rows_mine
    ev = [(eng.event_create(), eng.event_create()) for _ in range(max(args.steps, 20))]

    def step(i_timed=None):
        if preresampled:
            if i_timed is not None:
                eng.event_record(ev[i_timed][0])
            eng.calculate_cn(d_esa.ptr, d_fine.ptr, npix, K_G_II, out0)
            if i_timed is not None:
                eng.event_record(ev[i_timed][1])
            return
        eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
        whole = strip >= rows_mine
        if i_timed is not None and not whole:
            eng.event_record(ev[i_timed][0])
        for y0 in range(0, rows_mine, strip):
            rows = min(strip, rows_mine - y0)
            ptrs = [p + y0 * size if p else None for p in outs]
            if i_timed is not None and whole:
                # events carried by the dispatch itself: the kernel's own duration
                eng.time_next_strip(ev[i_timed][0], ev[i_timed][1])
            eng.cn_strip(d_esa.at(y0 * size), size, rows, d_cj.at(4 * y0), cond_mask, table_mask, ptrs)
        if i_timed is not None and not whole:
            eng.event_record(ev[i_timed][1])

    # The card needs some tens of milliseconds under load before its launch times settle after the idle
    # stretches of the set-up (allocations, the CPU baseline): the first 20-30 launches after a pause run
    # 1.5-5 % slower than the ones that follow (profiles/r02/time_series.jsonl, bench_ab_prepare_tile.json).
    # So the same step runs untimed for --pre-warm-ms before the W warm-up steps the contract asks for.
    pre_warm_steps = 0
    if args.pre_warm_ms > 0 and not fake_engine:
        t0 = time.monotonic()
        while (time.monotonic() - t0) * 1e3 < args.pre_warm_ms:
            for _ in range(10):
                step()
            eng.device_sync()
            pre_warm_steps += 10
    for _ in range(args.warmup):
        step()
    eng.device_sync()
    grp.barrier()
    eng.device_sync()
    t_start = time.monotonic()
    for i in range(args.steps):
        step(i)
    eng.device_sync()
    t_end = time.monotonic()
    grp.barrier()

    kernel_ms = [eng.elapsed_ms(a, b) for a, b in ev[:args.steps]]
    kname = "calculate_cn_kernel<true>" if preresampled else eng.last_kernel_name()
    launches = 1 if preresampled else (rows_mine + strip - 1) // strip
    avg_launch_s = float(np.mean(kernel_ms)) / 1e3 / launches
    if preresampled:
        alg_bytes = 3 * npix
    else:
        alg_bytes = gpu.strip_algorithmic_bytes(size, rows_mine, hs, hs, cond_mask, table_mask) / launches
    achieved = alg_bytes / avg_launch_s / 1e9

    def timed_launches(fn, n=20, warm=3):
        """`fn()` launches one kernel; its dispatch carries the events (gcn10_gpu_time_next_strip)."""
        for _ in range(warm):
            fn()
        eng.sync()
        for i in range(n):
            eng.time_next_strip(ev[i][0], ev[i][1])
            fn()
        eng.sync()
        return [eng.elapsed_ms(a, b) for a, b in ev[:n]]

    # Self-check (untimed, no oracle involved; straight after the timed region, before any other leg writes
    # into the raster): the bytes the timed launches left in the raster, against the same strip
    # computed once more by the most different variant the library has -- soil code BYTES instead of compact
    # words, one chunk per trip, no software pipeline, plain stores -- into another buffer.  A difference ends
    # the run: a fast kernel whose result depends on its launch shape is not a result.
    self_check = None
    if (not preresampled and not fake_engine and n_out == 1 and strip >= rows_mine and not args.no_self_check
            and hasattr(eng, "soil_words_state")):
        other = eng.alloc(npix)
        eng.memset(other.ptr, 0xEE, npix)
        words_state = None
        try:
            # out0 holds what the LAST TIMED launch wrote: nothing has touched it since the timed region
            words_state = eng.soil_words_state()
            for name, v in (("compact_soil", 0), ("ilp1", 1), ("prefetch", 0), ("nontemporal", 0)):
                eng.set_option(name, v)
            ptrs2 = [other.ptr if p else None for p in outs]
            eng.cn_strip(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask, ptrs2)
            eng.sync()
            k2 = eng.last_kernel_name()
        finally:
            eng.set_option("defaults", 0)
        differing = 0
        piece = 1 << 28
        for o in range(0, npix, piece):
            n = min(piece, npix - o)
            differing += int(np.count_nonzero(eng.download(out0 + o, (n,)) != eng.download(other.ptr + o, (n,))))
        other.close()
        if differing:
            raise SystemExit("bench.py: SELF-CHECK FAILED: %d of %d pixels differ between %s and %s" %
                             (differing, npix, kname, k2))
        self_check = {"pixels_compared": npix, "differing": 0, "timed_kernel": kname, "against": k2 +
                      ", soil code bytes, plain stores", "soil_words_state_of_the_timed_tile": words_state}
        # (the calibrated launch shape was dropped by the options above: nothing below uses the one-raster kernel)
        self_check["checked"] = "the raster as the last timed launch left it (no re-launch of the timed variant)"

    # diagnostic (GCN10_BENCH_AB=1): the same strip launch back to back, without gcn10_gpu_prepare_tile in between
    b2b = None
    if os.environ.get("GCN10_BENCH_AB") == "1" and not preresampled and strip >= rows_mine:
        ptrs0 = list(outs)
        b2b = {"timed_region_kernel_ms": [round(x, 4) for x in kernel_ms],
               "strip_only": _stats(timed_launches(
            lambda: eng.cn_strip(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask, ptrs0)))}

        # prepare_tile in front of every launch (timed_launches attaches the events to the NEXT strip dispatch)
        ms = []
        for i in range(20):
            eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
            eng.time_next_strip(ev[i][0], ev[i][1])
            eng.cn_strip(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask, ptrs0)
        eng.sync()
        b2b["after_prepare_tile"] = _stats([eng.elapsed_ms(a, b) for a, b in ev[:20]])
        ms = []
        for i in range(20):
            eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
            eng.sync()
            eng.time_next_strip(ev[i][0], ev[i][1])
            eng.cn_strip(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask, ptrs0)
            eng.sync()
        b2b["after_prepare_tile_and_sync"] = _stats([eng.elapsed_ms(a, b) for a, b in ev[:20]])
        b2b["strip_only_again"] = _stats(timed_launches(
            lambda: eng.cn_strip(d_esa.ptr, size, rows_mine, d_cj.ptr, cond_mask, table_mask, ptrs0)))

    # what this GPU streams when a kernel only moves the bytes (1 B read : 1 B written), same run
    copy = None
    if not args.no_also and not preresampled and hasattr(eng, "stream_copy"):
        nb = npix - npix % 16
        ms = timed_launches(lambda: eng.stream_copy(d_esa.ptr, out0, nb), warm=40)
        gbs = 2 * nb / float(np.mean(ms)) / 1e6
        copy = dict(_stats(ms), kernel="stream_copy_kernel", bytes_per_launch=2 * nb,
                    achieved_GBps=round(gbs, 1), frac_of_peak=round(gbs / HBM_PEAK_GBS, 4))

    also = None
    if want_also:
        # the product's per-block pass (18 rasters fused), outside the timed region, two ways:
        # B = buffers allocated at start-up, 3 warm-ups, 20 launches, events carried by each dispatch
        #     (the way `roofline` is measured);
        # A = the round-1 form of this leg: 17 fresh allocations right before, one warm-up, 5 launches
        #     between two separately recorded events.
        try:
            b18 = gpu.strip_algorithmic_bytes(size, size, hs, hs, 3, 0x1FF)
            all_outs = [out0] + [b.ptr for b in extra]
            eng.prepare_tile(d_coarse.ptr, hs, hs, d_ci.ptr, size)
            ms_b = timed_launches(lambda: eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, all_outs))
            k18 = eng.last_kernel_name()
            for b in extra:
                b.close()
            extra = [eng.alloc(npix) for _ in range(17)]
            all_outs = [out0] + [b.ptr for b in extra]
            e0, e1 = eng.event_create(), eng.event_create()
            eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, all_outs)
            eng.sync()
            eng.event_record(e0)
            for _ in range(5):
                eng.cn_strip(d_esa.ptr, size, size, d_cj.ptr, 3, 0x1FF, all_outs)
            eng.event_record(e1)
            eng.sync()
            ms_a = eng.elapsed_ms(e0, e1) / 5
            gbs = b18 / float(np.mean(ms_b)) / 1e6
            also = {"workload": "config4 (18 rasters fused, kernel only)", "kernel": k18,
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": traffic_from_profiles("config4") if (size == 36000 and args.pattern == "iid") else None,
                    "algorithmic_bytes_per_launch": int(b18),
                    "avg_launch_ms": _stats(ms_b)["avg_ms"], "median_launch_ms": _stats(ms_b)["median_ms"],
                    "min_launch_ms": _stats(ms_b)["min_ms"],
                    "cn_gpx_per_s": round(npix * 18 / float(np.mean(ms_b)) / 1e6, 2),
                    "round1_style_ms_per_launch": round(ms_a, 4),
                    "round1_style_note": "17 fresh 1.3 GB allocations right before, 1 warm-up, 5 launches "
                                         "between two separately recorded events"}
        except Exception as exc:       # never lose the bench line over the extra leg
            also = {"error": str(exc)}
        finally:
            for b in extra:
                b.close()

    uncalibrated_frac = None
    try:
        tried_ms = (placement or {}).get("allocations_tried_best_ms") if isinstance(placement, dict) else None
        if tried_ms:
            uncalibrated_frac = round(alg_bytes / (float(np.median(tried_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    except Exception:
        uncalibrated_frac = None
    mine = {"rank": rank, "device": device, "pci_bus_id": bus, "numa_node": numa, "cpus_bound": bound,
            "t_start": t_start, "t_end": t_end, "elapsed_s": round(t_end - t_start, 6),
            "kernel_avg_ms": round(avg_launch_s * 1e3, 4), "kernel": kname, "rows": [band[0], band[1]],
            "copy_GBps": copy["achieved_GBps"] if copy else None}
    ranks = grp.all_gather(mine)
    if rank == 0:
        t0 = min(r["t_start"] for r in ranks)
        elapsed = max(r["t_end"] for r in ranks) - t0        # whole job: first start to last end
        for r in ranks:
            r["start_offset_ms"] = round((r.pop("t_start") - t0) * 1e3, 3)
            r["end_offset_ms"] = round((r.pop("t_end") - t0) * 1e3, 3)
        total_px = sum(size * (r["rows"][1] - r["rows"][0]) for r in ranks)      # N blocks (weak) or one block (strong)
        value = total_px * n_out / elapsed * args.steps / 1e9
        line = {
            "metric": "CN Gpixels/sec", "value": round(value, 3), "unit": "Gpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if not fake_engine else "FAKE ENGINE (launcher test, not a measurement)",
            "config": {"workload": "%s: one %dx%d uint8 landcover block %s, %s, %d CN raster(s) per step; "
                                   "soil %s" % (args.workload, size, size,
                                                "split into row bands over the GPUs" if strong else "per GPU",
                                                "lookup g_ii (ARC-II)" if n_out < 9 else "all 9 lookups",
                                                n_out,
                                                "pre-resampled full-res tile" if preresampled
                                                else "window %dx%d resampled in-kernel" % (hs, hs)),
                       "pattern": args.pattern, "strip_rows": strip, "blocks_per_step_per_gpu": 1,
                       "device": info["name"], "cus": info["cus"],
                       "rank_sync": "none" if world == 1 else grp.backend,
                       "pre_warm_steps_untimed": pre_warm_steps},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the committed PMC passes were taken at the default shape only
                         "traffic": traffic_from_profiles(args.workload, kernel=kname)
                         if (size == 36000 and strip == size and args.pattern == "iid") else None,
                         "traffic_kernel": traffic_from_profiles(args.workload, "kernel", kernel=kname)
                         if (size == 36000 and strip == size and args.pattern == "iid") else None,
                         "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes, not this run)",
                         # what ONE ordinary allocation gets: the median over the candidate rasters the untimed
                         # calibration tried (bin/gcn10 does not calibrate: it never writes a raster to HBM)
                         "uncalibrated_frac": uncalibrated_frac,
                         "kernel": kname, "algorithmic_bytes_per_launch": int(alg_bytes),
                         "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                         "median_launch_ms": round(float(np.median(kernel_ms)) / launches, 4),
                         "min_launch_ms": round(float(np.min(kernel_ms)) / launches, 4),
                         "copy_ceiling": copy,          # the plain copy landcover -> the same raster buffer
                         "placement": placement,        # gcn10_gpu_tune_single_raster's report (setup, untimed)
                         "frac_of_copy": round(achieved / copy["achieved_GBps"], 4) if copy else None},
            "cpu_baseline": cpu,
            "per_rank": ranks,
        }
        if self_check:
            line["self_check"] = self_check
        if b2b:
            line["ab_prepare_tile"] = b2b
        if also:
            line["also"] = also
        print(json.dumps(line))
        sys.stdout.flush()
    grp.close()
    eng.close()


if __name__ == "__main__":
    main()

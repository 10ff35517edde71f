"""Times the oracle's reference-shaped block pass on host cores (bench.py's cpu_baseline leg).

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Runs in its own process tree (never
touches a GPU): P worker processes -- the stand-in of `mpirun -n P`
(src/main.c:171) -- each run oracle_process_block_subset() (same loops and
per-raster malloc/memcpy/memset as src/cn.c:218-290, I/O removed) on its own
strip of a synthetic block.  SURVEY.md 8(d): P = 1 and P = all physical cores
(here also P = 16, the CPU share of a one-GPU box), every worker pinned to one
physical core; then, as a "best CPU" line, the fused single pass of
cn_fused_cpu.c built -march=native on this host, on all physical cores.
Prints one JSON object.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def physical_cores():
    """One logical CPU of every physical core this process may run on: [cpu, ...]."""
    allowed = os.sched_getaffinity(0)
    seen, cpus = set(), []
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if not line.strip():
                    if "processor" in cur:
                        cpu = int(cur["processor"])
                        key = (cur.get("physical id", "0"), cur.get("core id", str(cpu)))
                        if cpu in allowed and key not in seen:
                            seen.add(key)
                            cpus.append(cpu)
                    cur = {}
                    continue
                k, _, v = line.partition(":")
                cur[k.strip()] = v.strip()
    except OSError:
        pass
    return cpus or sorted(allowed)


def cpu_quota_cores():
    """CPU bandwidth limit of this cgroup in cores (cpu.max), or None when unlimited / unknown."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] == "max":
                    return None
                return float(parts[0]) / float(parts[1])
            q = float(parts[0])
            if q <= 0:
                return None
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                return q / float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            continue
    return None


def _worker(args):
    import numpy as np
    from oracle import cn_oracle_c as oc
    from oracle import cn_oracle_np as onp
    seed, W, rows, cond_mask, table_mask, lookups, cpu, fused = args
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    tables = np.stack([oc.load_lookup_table(os.path.join(lookups, "default_lookup_%s_%s.csv" % (hc, arc)))[0]
                       for hc in onp.HCS for arc in onp.ARCS])
    rng = np.random.default_rng(seed)
    classes = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], dtype=np.uint8)
    esa = classes[rng.integers(0, 12, size=(rows, W), dtype=np.uint8)]
    hsy = max(1, rows // 25)
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], np.uint8), size=(hsy, 1440))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [0.0, 3.0 / 1440, 0.0, 3.0, 0.0, -(rows * 3.0 / W) / hsy]
    if fused:
        # the fused pass writes its rasters into buffers of the caller (the reference-shaped pass
        # allocates, fills and frees its own per raster, src/cn.c:264-290,376-377)
        outs = [np.zeros((rows, W), dtype=np.uint8)
                if (cond_mask >> (i // 9)) & 1 and (table_mask >> (i % 9)) & 1 else None for i in range(18)]
        t0 = time.perf_counter()
        oc.fused_block(esa, gt, coarse, sgt, tables, cond_mask=cond_mask, table_mask=table_mask,
                       want_output=False, _out=outs)
        return time.perf_counter() - t0
    t0 = time.perf_counter()
    oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=cond_mask, table_mask=table_mask,
                         want_output=False)
    return time.perf_counter() - t0


def run_leg(label, p, cpus, W, rows, cond_mask, table_mask, lookups, fused=False):
    n_out = bin(cond_mask & 3).count("1") * bin(table_mask & 0x1FF).count("1")
    jobs = [(1000 + i, W, rows, cond_mask, table_mask, lookups, cpus[i % len(cpus)] if cpus else None, fused)
            for i in range(p)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(p) as pool:
        times = pool.map(_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    # rate over the compute part only (input generation excluded): all workers run
    # concurrently, the slowest defines the job
    return {"label": label, "procs": p, "rows_per_proc": rows, "gpx_per_s": p * W * rows * n_out / max(times) / 1e9,
            "worker_seconds_max": round(max(times), 3), "worker_seconds_min": round(min(times), 3),
            "wall_seconds": round(wall, 2), "pinned": bool(cpus)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=36000)
    ap.add_argument("--rows", type=int, default=0, help="rows per worker of the P = 1 and P = 16 legs (0 = by raster count)")
    ap.add_argument("--cond-mask", type=int, default=3)
    ap.add_argument("--table-mask", type=int, default=0x1FF)
    ap.add_argument("--share", type=int, default=int(os.environ.get("GCN10_CPU_BASELINE_PROCS", "16")),
                    help="the middle leg: the CPU share of a one-GPU box")
    ap.add_argument("--lookups", default=os.path.join(ROOT, "tests", "golden", "lookups"))
    ap.add_argument("--no-best", action="store_true")
    a = ap.parse_args()
    from oracle import cn_oracle_c as oc
    oc.build()
    if not a.no_best:
        oc.build_native(force=True)
    ncores = len(os.sched_getaffinity(0))
    cpus = physical_cores()
    nphys = len(cpus)
    n_out = bin(a.cond_mask & 3).count("1") * bin(a.table_mask & 0x1FF).count("1")
    W = a.width
    # a few seconds of CPU work per worker: 12000 rows for one raster, 36000/n rows for n
    rows = a.rows or max(16, min(W, 12000 if n_out == 1 else 36000 // n_out))
    # all-cores leg: shorter strips so that P workers x (landcover + resampled + adjusted + raster,
    # src/cn.c:209,264,278) stay far below the box's memory: 4 x rows x W x P bytes
    mem_cap = 64e9
    rows_all = int(max(16, min(rows, mem_cap / (4.0 * W * max(nphys, 1)))))
    runs = [run_leg("P=1", 1, cpus, W, rows, a.cond_mask, a.table_mask, a.lookups)]
    share = max(1, min(a.share, nphys))
    if share > 1:
        runs.append(run_leg("P=%d (one-GPU box share)" % share, share, cpus, W, rows, a.cond_mask, a.table_mask, a.lookups))
    if nphys > share:
        runs.append(run_leg("P=%d (all physical cores)" % nphys, nphys, cpus, W, rows_all, a.cond_mask, a.table_mask,
                            a.lookups))
    best = None
    if not a.no_best:
        rows_best = int(max(16, min(W, mem_cap / ((1.0 + n_out) * W * max(nphys, 1)), 4 * rows_all)))
        best = run_leg("fused single pass, -O3 -march=native, P=%d" % nphys, nphys, cpus, W, rows_best, a.cond_mask,
                       a.table_mask, a.lookups, fused=True)
        best["gpx_per_s"] = round(best["gpx_per_s"], 4)
    # the figure quoted as cpu_baseline.value: the fastest reference-shaped run (all physical cores where
    # the box lets this process use them; a cgroup CPU quota below that -- cpu_quota_cores -- makes
    # the quota-sized run the faster one)
    headline = max(runs, key=lambda r: r["gpx_per_s"])
    for r in runs:
        r["gpx_per_s"] = round(r["gpx_per_s"], 4)
    print(json.dumps({"cores_available": ncores, "physical_cores": nphys, "cpu_quota_cores": cpu_quota_cores(),
                      "n_out": n_out, "runs": runs, "headline": headline, "best_cpu": best}))


if __name__ == "__main__":
    main()

"""Times the oracle's reference-shaped block pass on host cores (bench.py's cpu_baseline leg).

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Runs in its own process tree (never
touches a GPU): P worker processes -- the stand-in of `mpirun -n P`
(src/main.c:171) -- each run oracle_process_block_subset() (same loops and
per-raster malloc/memcpy/memset as src/cn.c:218-290, I/O removed) on its own
strip of a synthetic block.  Prints one JSON object.
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


def _worker(args):
    import numpy as np
    from oracle import cn_oracle_c as oc
    seed, W, rows, cond_mask, table_mask, lookups = args
    from oracle import cn_oracle_np as onp
    tables = np.stack([oc.load_lookup_table(os.path.join(lookups, "default_lookup_%s_%s.csv" % (hc, arc)))[0]
                       for hc in onp.HCS for arc in onp.ARCS])
    rng = np.random.default_rng(seed)
    classes = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], dtype=np.uint8)
    esa = classes[rng.integers(0, 12, size=(rows, W), dtype=np.uint8)]
    hsy = max(1, rows // 25)
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], np.uint8), size=(hsy, 1440))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [0.0, 3.0 / 1440, 0.0, 3.0, 0.0, -(rows * 3.0 / W) / hsy]
    t0 = time.perf_counter()
    oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=cond_mask, table_mask=table_mask,
                         want_output=False)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=36000)
    ap.add_argument("--rows", type=int, default=2000)
    ap.add_argument("--cond-mask", type=int, default=3)
    ap.add_argument("--table-mask", type=int, default=0x1FF)
    ap.add_argument("--procs", type=int, default=0, help="0 = all cores this process may use")
    ap.add_argument("--lookups", default=os.path.join(ROOT, "tests", "golden", "lookups"))
    a = ap.parse_args()
    from oracle import cn_oracle_c as oc
    oc.build()
    ncores = len(os.sched_getaffinity(0))
    # a one-GPU box's CPU share is 16 cores; more workers than that would time
    # other tenants' cores (and 256 x 4 rasters in flight is a lot of RAM)
    procs = a.procs or min(ncores, int(os.environ.get("GCN10_CPU_BASELINE_PROCS", "16")))
    n_out = bin(a.cond_mask & 3).count("1") * bin(a.table_mask & 0x1FF).count("1")
    px = a.width * a.rows
    res = {}
    for label, p in (("single", 1), ("multi", procs)):
        if label == "multi" and p == 1:
            res[label] = res["single"]
            continue
        jobs = [(1000 + i, a.width, a.rows, a.cond_mask, a.table_mask, a.lookups) for i in range(p)]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(p) as pool:
            times = pool.map(_worker, jobs)
        wall = time.perf_counter() - t0
        # rate over the compute part only (input generation excluded): all
        # workers run concurrently, the slowest defines the job
        res[label] = {"procs": p, "gpx_per_s": p * px * n_out / max(times) / 1e9,
                      "worker_seconds_max": max(times), "wall_seconds": wall}
    print(json.dumps({"cores_available": ncores, "n_out": n_out, "sample_px_per_proc": px,
                      "single": res["single"], "multi": res["multi"]}))


if __name__ == "__main__":
    main()

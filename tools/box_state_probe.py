#!/usr/bin/env python3
"""What differs between a "fast" and a "slow" process on the same kind of box?  (GPU only; diagnostic.)

Prints, as one JSON object: the DPM clock tables sysfs shows (sclk / mclk / fclk / socclk, the active level
starred) sampled before, DURING a 2-second loop of stream copies and after it; the power cap / perf level files;
VRAM use; and the dispatch-timed plain copy over several allocations.  Run it as the FIRST GPU process of a
gpurun call and again after the test-suite to see which of them moves with the copy time."""
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from gcn10_amd import gpu  # noqa: E402

FILES = ["pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level",
         "mem_info_vram_used", "mem_info_vram_total", "mem_info_vis_vram_used", "gpu_busy_percent",
         "mem_busy_percent", "current_compute_partition", "current_memory_partition"]


def cards():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or os.path.exists(os.path.join(d, "mem_info_vram_total")):
            out.append(d)
    return out


ONLY = None     # sysfs device directory of the card this process computes on, once known


def snapshot():
    snap = {}
    for d in ([ONLY] if ONLY else cards()):
        one = {}
        for f in FILES:
            p = os.path.join(d, f)
            try:
                with open(p) as fh:
                    one[f] = fh.read().strip().replace("\n", " | ")
            except OSError as e:
                one[f] = "n/a (%s)" % e.__class__.__name__
        for h in glob.glob(os.path.join(d, "hwmon/hwmon*")):
            for f in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input",
                      "temp2_input", "temp3_input"):
                p = os.path.join(h, f)
                if os.path.exists(p):
                    try:
                        with open(p) as fh:
                            one["hwmon_" + f] = fh.read().strip()
                    except OSError:
                        pass
        snap[d.split("/")[4] if d.startswith("/sys/class/drm/") else "mine"] = one
    return snap


def main():
    size = int(os.environ.get("PROBE_SIZE", "36000"))
    npix = size * size
    nb = npix - npix % 16
    out = {"label": os.environ.get("PROBE_LABEL", ""), "before": snapshot()}
    eng = gpu.Engine(0)
    out["pci"] = eng.pci_bus_id()
    global ONLY
    mine = "/sys/bus/pci/devices/%s" % out["pci"].lower()
    if os.path.exists(os.path.join(mine, "pp_dpm_sclk")):
        ONLY = mine
        out["before"] = {k: v for k, v in out["before"].items()
                         if os.path.realpath("/sys/class/drm/%s/device" % k) == os.path.realpath(mine)}
    rng = np.random.default_rng(1)
    src = eng.upload(rng.integers(0, 256, npix, dtype=np.uint8))
    dsts = [eng.alloc(npix) for _ in range(int(os.environ.get("PROBE_ALLOCS", "6")))]
    ev = [(eng.event_create(), eng.event_create()) for _ in range(5)]

    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(snapshot())
            time.sleep(0.25)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.0:
        for _ in range(50):
            eng.stream_copy(src.ptr, dsts[0].ptr, nb)
        eng.sync()
        n += 50
    stop.set()
    th.join()
    out["copies_in_loop"] = n
    out["loop_ms_per_copy_wall"] = round((time.time() - t0) * 1e3 / n, 4)
    # keep the distinct readings only
    seen = []
    for s in samples:
        if s not in seen:
            seen.append(s)
    out["during"] = seen[:6]
    res = []
    for rnd in range(3):
        for j, d in enumerate(dsts):
            eng.stream_copy(src.ptr, d.ptr, nb)
            for k in range(5):
                eng.time_next_strip(*ev[k])
                eng.stream_copy(src.ptr, d.ptr, nb)
            eng.sync()
            ms = sorted(eng.elapsed_ms(*ev[k]) for k in range(5))[2]
            if rnd == 0:
                res.append([ms])
            else:
                res[j].append(ms)
    out["copy_ms_by_allocation"] = [[round(v, 4) for v in r] for r in res]
    out["dst_ptrs"] = [hex(d.ptr) for d in dsts]
    out["src_ptr"] = hex(src.ptr)
    out["after"] = snapshot()
    if os.environ.get("PROBE_COMPACT"):
        def brief(sn):
            v = next(iter(sn.values())) if sn else {}
            return {k: v.get(k) for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "hwmon_power1_input",
                                           "hwmon_temp2_input", "hwmon_temp3_input", "mem_info_vram_used",
                                           "gpu_busy_percent", "mem_busy_percent")}
        out = {"label": out["label"], "pci": out["pci"], "copy_ms_by_allocation": [r[1] for r in out["copy_ms_by_allocation"]],
               "before": brief(out["before"]), "during": [brief(x) for x in out["during"][1:3]], "after": brief(out["after"])}
        print(json.dumps(out))
    else:
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

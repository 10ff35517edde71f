#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profiles/run_profiles_r03.sh into the small files kept under profiles/r03/.
Round 3: the PMC traffic is also kept PER KERNEL VARIANT (the calibration launches several instantiations of the
single-raster kernel on the full block: `config2_by_kernel`), so that bench.py can quote the traffic of the very
variant it times (`roofline.traffic_kernel`), whatever the calibration chooses on the box at hand."""
import collections
import csv
import glob
import json
import os
import re
import sys

O, P, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.makedirs(P, exist_ok=True)


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def short(name):
    m = re.search(r"(cn_strip_kernel<[^>]*>|stream_copy_kernel|expand_x_codes<[^>]*>)", name)
    return m.group(1) if m else name[:60]


def bench_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def timed_dispatches(rows, name_key, start_key, chosen, k):
    """rows of one trace: the dispatches that bench.py timed, per kernel."""
    rows = sorted(rows, key=lambda r: int(r[start_key]))
    names = [short(r[name_key]) for r in rows]
    first_copy = names.index("stream_copy_kernel") if "stream_copy_kernel" in names else len(rows)
    strip = [r for r, n in zip(rows[:first_copy], names[:first_copy]) if n == chosen][-k:]
    copy = [r for r, n in zip(rows, names) if n == "stream_copy_kernel"][-20:]
    k18 = [r for r, n in zip(rows, names) if n.startswith("cn_strip_kernel<0, 3")]
    k18 = k18[3:23] if len(k18) >= 23 else k18        # 3 warm-ups, 20 timed, then the round-1-style 6
    return {"strip": strip, "copy": copy, "config4": k18}


line = bench_line(os.path.join(O, "kt.json"))
chosen = line["roofline"]["kernel"]
kt = newest(os.path.join(O, "kt", "**", "*kernel_trace.csv"))
rows = list(csv.DictReader(open(kt)))
sel = timed_dispatches(rows, "Kernel_Name", "Start_Timestamp", chosen, K)
summary = {"bench_line_under_rocprof": {"kernel": chosen, "avg_launch_ms": line["roofline"]["avg_launch_ms"],
                                         "frac": line["roofline"]["frac"], "placement": line["roofline"].get("placement"),
                                         "copy_ceiling": line["roofline"].get("copy_ceiling"),
                                         "also": line.get("also")}}
for key, rs in sel.items():
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs]
    if d:
        summary[key] = {"kernel": short(rs[0]["Kernel_Name"]), "dispatches": len(d), "avg_us": round(sum(d) / len(d), 2),
                        "min_us": round(min(d), 2), "max_us": round(max(d), 2),
                        "vgpr": rs[0].get("VGPR_Count") or rs[0].get("Arch_VGPR_Count"), "sgpr": rs[0].get("SGPR_Count"),
                        "grid": rs[0].get("Grid_Size") or rs[0].get("Grid_Size_X"), "lds": rs[0].get("LDS_Block_Size")}
# whole-run per-kernel stats as rocprofv3 prints them (includes the calibration's launches)
ks = newest(os.path.join(O, "kt", "**", "*kernel_stats.csv"))
open(os.path.join(P, "kernel_stats_bench_config2.csv"), "w").write(open(ks).read())

traffic = {}
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = newest(os.path.join(O, d, "**", "*counter_collection.csv"))
    chosen_d = bench_line(os.path.join(O, d + ".json"))["roofline"]["kernel"]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    order = sorted(per)
    names = [k[1] for k in order]
    first_copy = names.index("stream_copy_kernel") if "stream_copy_kernel" in names else len(order)
    strip = [per[k] for k in order[:first_copy] if k[1] == chosen_d][-4:]
    copy = [per[k] for k in order if k[1] == "stream_copy_kernel"][-20:]
    k18 = [per[k] for k in order if k[1].startswith("cn_strip_kernel<0, 3")]
    # every single-raster variant the calibration launched (full-block launches all): average per variant
    by_var = collections.defaultdict(list)
    for k in order:
        if k[1].startswith("cn_strip_kernel<1,"):
            by_var[k[1]].append(per[k])
    for kn, vals in by_var.items():
        t = traffic.setdefault("config2_by_kernel", {}).setdefault(kn, {})
        t[counter + "_KiB"] = sum(vals) / len(vals)
        t[counter + "_dispatches"] = len(vals)
    for key, vals, kn in (("config2", strip, chosen_d), ("copy", copy, "stream_copy_kernel"),
                          ("config4", k18, next((n for n in names if n.startswith("cn_strip_kernel<0, 3")), ""))):
        if vals:
            t = traffic.setdefault(key, {"kernel": kn})
            t[counter + "_KiB"] = sum(vals) / len(vals)
            t[counter + "_dispatches"] = len(vals)
for kn, t in traffic.get("config2_by_kernel", {}).items():
    if "FETCH_SIZE_KiB" in t and "WRITE_SIZE_KiB" in t:
        t["hbm_bytes_per_launch"] = int((2 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024)
for key, t in traffic.items():
    if key == "config2_by_kernel":
        continue
    if "FETCH_SIZE_KiB" in t and "WRITE_SIZE_KiB" in t:
        t["hbm_bytes_per_launch"] = int((2 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024)
        t["correction"] = ("gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled; WRITE_SIZE exact for "
                           "16-B/lane stores; both in KiB (MI355X_MICROARCH.md, HBM section)")
summary["pmc_traffic"] = traffic
json.dump(traffic, open(os.path.join(os.path.dirname(P.rstrip("/")), "pmc_traffic.json"), "w"), indent=1)

sq = {}
for d in ("sq1", "sq2"):
    f = newest(os.path.join(O, d, "**", "*counter_collection.csv"))
    chosen_d = bench_line(os.path.join(O, d + ".json"))["roofline"]["kernel"]
    per = collections.defaultdict(lambda: collections.OrderedDict())
    for r in csv.DictReader(open(f)):
        per[r["Counter_Name"]].setdefault((int(r["Dispatch_Id"]), short(r["Kernel_Name"])), 0.0)
        per[r["Counter_Name"]][(int(r["Dispatch_Id"]), short(r["Kernel_Name"]))] += float(r["Counter_Value"])
    for counter, vals in per.items():
        order = sorted(vals)
        names = [k[1] for k in order]
        first_copy = names.index("stream_copy_kernel") if "stream_copy_kernel" in names else len(order)
        strip = [vals[k] for k in order[:first_copy] if k[1] == chosen_d][-4:]
        k18 = [vals[k] for k in order if k[1].startswith("cn_strip_kernel<0, 3")]
        cp = [vals[k] for k in order if k[1] == "stream_copy_kernel"][-20:]
        for key, v in (("config2 " + chosen_d, strip), ("config4", k18), ("copy", cp)):
            if v:
                sq.setdefault(key, {})[counter] = round(sum(v) / len(v), 1)
summary["sq_counters_per_launch"] = sq
json.dump(summary, open(os.path.join(P, "profile_summary.json"), "w"), indent=1)
open(os.path.join(P, "bench_final.json"), "w").write(open(os.path.join(O, "bench_final.json")).read())
print(json.dumps({k: summary[k] for k in ("strip", "copy", "config4") if k in summary}))
print(json.dumps(traffic)[:600])

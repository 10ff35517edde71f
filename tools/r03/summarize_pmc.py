#!/usr/bin/env python3
"""Per-kernel sums of the SQ counter passes of tools/r03/run_baseline.sh (rocprofv3 --pmc CSVs):
one row per kernel and counter, summed over all dispatches of the run, plus dispatch counts.
Quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) are left in their own unit."""
import collections
import csv
import glob
import json
import sys


def short(name):
    for key in ("inflate_kernel", "untile_kernel", "fused_stats_kernel", "fused_emit_kernel", "deflate_codes_wave_kernel",
                "expand_x_codes", "predictor", "fused_tokens_kernel", "fused_raster_stats_kernel"):
        if key in name:
            return key
    return name[:48]


def main():
    root = sys.argv[1]
    out = collections.OrderedDict()
    for f in sorted(glob.glob(root + "/pmc*_natural/*/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            row = out.setdefault(k, {})
            row["dispatches"] = len(calls[k])
            for a, b in v.items():
                row[a] = round(b)
    for k, v in out.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()

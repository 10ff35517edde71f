#!/usr/bin/env python3
"""Kernel-level timing of the fused tile encoder on one 1024-row strip of a 36000-px block, with
the timing-experiment switches of gcn10_gpu_set_option("fused_diag").  One JSON line."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gcn10_amd import gpu, host  # noqa: E402
LOOKUPS = os.path.join(ROOT, "tests", "golden", "lookups")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pattern", default="natural")
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--diags", default="0,1,2,6")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--codes-stops", default="0", help='pass B leaves after phase (value - 1) (timing only; 0 = complete)')
    ap.add_argument("--stats-stops", default="0", help='pass F-A leaves after phase (value - 1) (timing only; 0 = complete)')
    ap.add_argument("--emits", default="1", help='pass F-C forms to time: 0 = lock step, 1 = every wave its own quarter')
    ap.add_argument("--parses", default="1", help='pass F-A forms to time: 0 = one lane per row, 1 = one lane per 64-px segment')
    a = ap.parse_args()
    W, H = 36000, a.rows
    esa, _, _, _ = bench.synth_block(1, 4096, a.pattern)
    esa = np.ascontiguousarray(np.tile(esa[:min(H, 4096)], ((H + 4095) // 4096, 9))[:H, :W])
    rng = np.random.default_rng(2)
    hsx, hsy = W // 25, max(H // 25, 1)
    coarse = rng.choice(bench.HSG_CODES, size=(hsy, hsx)).astype(np.uint8)
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [0.0, 3.0 / hsx, 0.0, 3.0, 0.0, -3.0 / hsx]
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    tabs = host.load_all_lookup_tables(LOOKUPS)
    res = {"pattern": a.pattern, "rows": H, "W": W, "ms": {}}
    with gpu.Engine(0) as e:
        e.set_tables(tabs)
        bufs = [e.upload(x) for x in (esa, coarse, ci, cj)]
        e.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
        n = 18
        across, down = (W + 255) // 256, (H + 255) // 256
        cap = int(gpu.lib().gcn10_gpu_deflate_arena_bound(W, H, n))
        arena, table, cursor = e.alloc(cap), e.alloc(n * across * down * 8), e.alloc(8)
        e0, e1 = e.event_create(), e.event_create()
        for parse, emit, d, stop, cstop in [(int(p_), int(e_), int(v), int(st), int(cs)) for p_ in a.parses.split(",")
                                            for e_ in a.emits.split(",") for v in a.diags.split(",")
                                            for st in a.stats_stops.split(",") for cs in a.codes_stops.split(",")]:
            e.set_option("fused_stats_stop", stop)
            e.set_option("codes_stop", cstop)
            e.set_option("fused_parse", parse)
            e.set_option("fused_emit", emit)
            e.set_option("fused_diag", d)
            ms = []
            for rep in range(a.reps + 1):
                e.event_record(e0)
                e._chk(gpu.lib().gcn10_gpu_deflate_fused_strip(e._ctx, bufs[0].ptr, W, H, bufs[3].ptr, 3, 0x1FF,
                                                               arena.ptr, cap, table.ptr, cursor.ptr, None), "fused")
                e.event_record(e1)
                e.event_sync(e1)
                ms.append(e.elapsed_ms(e0, e1))
            used = int(e.download(cursor.ptr, (1,), dtype=np.uint64)[0])
            res["ms"]["parse%d_emit%d_diag%d%s%s" % (parse, emit, d, "_stop%d" % stop if stop else "",
                                                     "_cstop%d" % cstop if cstop else "")] = round(min(ms[1:]), 3)
            if d == 0 and stop == 0 and cstop == 0:
                res["arena_bytes_parse%d_emit%d" % (parse, emit)] = used
        e.set_option("defaults", 0)
        for b in bufs + [arena, table, cursor]:
            b.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised check of the fused tile encoder (not collected by pytest; run it on a GPU box:
`python tests/fuzz_fused_encoder.py --seconds 120`).  Random block shapes, landcover of several
textures, random soil windows and geotransforms, the shipped tables or random ones with few live
classes, random raster subsets: every stream of every raster must inflate (stock zlib) to the
oracle's tile.  Exit code 1 on any difference."""
import argparse
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn10_amd import gpu, host  # noqa: E402
from oracle import cn_oracle_c as oc  # noqa: E402
from oracle import cn_oracle_np as onp  # noqa: E402
from tests.util import ESA_NASTY, HSG_NASTY, random_tables  # noqa: E402

LOOKUPS = os.path.join(ROOT, "tests", "golden", "lookups")


def landcover(rng, H, W, values):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return rng.choice(values, size=(H, W)).astype(np.uint8)
    if kind == 1:
        return np.full((H, W), rng.choice(values), np.uint8)
    s = int(rng.integers(2, 80))
    small = rng.choice(values, size=((H + s - 1) // s, (W + s - 1) // s)).astype(np.uint8)
    img = np.repeat(np.repeat(small, s, axis=0), s, axis=1)[:H, :W]
    if kind >= 3:
        flip = rng.random((H, W)) < float(rng.uniform(0.001, 0.2))
        img = np.where(flip, rng.choice(values, size=(H, W)), img)
    return np.ascontiguousarray(img, dtype=np.uint8)


class _Args:
    pass


def run(seed=1, seconds=None, cases=None, verbose=True):
    """Runs until `cases` random blocks are done (or `seconds` have passed).  Returns
    (cases, streams checked, first difference or None)."""
    a = _Args()
    a.seed, a.seconds = seed, (seconds if seconds is not None else 1e9)
    max_cases = cases if cases is not None else 1 << 60
    rng = np.random.default_rng(a.seed)
    shipped = np.stack([oc.load_lookup_table(os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc)))[0]
                        for hc in onp.HCS for arc in onp.ARCS])
    t_end = time.time() + a.seconds
    t_print = time.time()
    cases = tiles_checked = 0
    with gpu.Engine(0) as e:
        while time.time() < t_end and cases < max_cases:
            H = int(rng.integers(1, 900))
            W = int(rng.integers(1, 1400))
            if rng.random() < 0.5:
                tabs, values = shipped, ESA_NASTY
            else:
                tabs = random_tables(int(rng.integers(0, 1 << 30)), 9)
                live = int(rng.integers(1, 7))
                tabs[:, live:, :] = 255                 # few live classes: at most 256 pixel classes
                values = np.arange(0, live + 3, dtype=np.uint8)
            esa = landcover(rng, H, W, values)
            ratio = float(rng.uniform(5.0, 40.0))
            hsy, hsx = int(H / ratio) + 2, int(W / ratio) + 2
            coarse = rng.choice(HSG_NASTY if rng.random() < 0.7 else np.array([1, 2, 3, 4], np.uint8),
                                size=(hsy, hsx)).astype(np.uint8)
            px = float(rng.uniform(1e-4, 1e-2))
            gt = [float(rng.uniform(-180, 170)), px, 0.0, float(rng.uniform(-80, 80)), 0.0, -px]
            sgt = [gt[0] - float(rng.uniform(0, 1)) * px * ratio, px * ratio, 0.0,
                   gt[3] + float(rng.uniform(0, 1)) * px * ratio, 0.0, -px * ratio]
            cond_mask = int(rng.integers(1, 4))
            table_mask = int(rng.integers(1, 512))
            ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
            e.set_tables(tabs)
            bufs = [e.upload(x) for x in (esa, coarse, ci, cj)]
            e.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
            data, table, used = e.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr, cond_mask, table_mask)
            for b in bufs:
                b.close()
            want = oc.process_block_mem(esa, gt, coarse, sgt, tabs, cond_mask=cond_mask, table_mask=table_mask)
            sel = [r for r in range(18) if (cond_mask >> (r // 9)) & 1 and (table_mask >> (r % 9)) & 1]
            for j, r in enumerate(sel):
                for ty in range((H + 255) // 256):
                    for tx in range((W + 255) // 256):
                        off, size = int(table[j, ty, tx, 0]), int(table[j, ty, tx, 1])
                        exp = np.zeros((256, 256), np.uint8)
                        part = want[r][ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                        exp[:part.shape[0], :part.shape[1]] = part
                        try:
                            ok = off + size <= used and zlib.decompress(data[off:off + size].tobytes()) == exp.tobytes()
                        except zlib.error:
                            ok = False
                        if not ok:
                            return cases, tiles_checked, "case %d seed %d H %d W %d raster %d tile (%d, %d)" % (
                                cases, a.seed, H, W, r, ty, tx)
                        tiles_checked += 1
            cases += 1
            if verbose and time.time() - t_print > 60:
                t_print = time.time()
                print("... %d cases, %d streams so far, no difference" % (cases, tiles_checked), flush=True)
    return cases, tiles_checked, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=None)
    ap.add_argument("--cases", type=int, default=None, help="fixed budget of random blocks (instead of a time)")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if a.seconds is None and a.cases is None:
        a.seconds = 120.0
    n, streams, diff = run(a.seed, a.seconds, a.cases)
    if diff:
        print("DIFFERENCE: " + diff)
        sys.exit(1)
    print("seed %d: cases %d, streams checked %d, differences 0" % (a.seed, n, streams))


if __name__ == "__main__":
    main()

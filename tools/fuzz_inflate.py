#!/usr/bin/env python3
"""Randomised check of gcn10_gpu_inflate_tiles against stock zlib: many streams per launch, every
level / strategy / window size / flush pattern, data from several generators, plus corrupted
copies (which must either be refused or decode exactly as zlib decodes them).  Prints a summary;
exit code 1 on any difference."""
import argparse
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn10_amd import gpu  # noqa: E402


def gen(rng, n):
    a = gen_any(rng, n)
    return a if len(a) == n else np.resize(a, n)


def gen_any(rng, n):
    kind = int(rng.integers(0, 9))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.zeros(n, np.uint8) + np.uint8(rng.integers(0, 256))
    if kind == 2:       # runs of a few symbols
        k = int(rng.integers(2, 40))
        runs = rng.geometric(float(rng.uniform(0.01, 0.6)), size=n // 2 + 8)
        return np.repeat(rng.integers(0, k, len(runs)).astype(np.uint8), runs)[:n].copy()
    if kind == 3:       # very skewed: long Huffman codes
        p = float(rng.uniform(0.15, 0.7))
        return np.minimum(rng.geometric(p, n) - 1, 255).astype(np.uint8)
    if kind == 4:       # periodic, self-overlapping copies
        out = []
        total = 0
        while total < n:
            per = int(rng.integers(1, 300))
            rep = int(rng.integers(2, 60))
            out.append(np.tile(rng.integers(0, 256, per, dtype=np.uint8), rep))
            total += per * rep
        return np.concatenate(out)[:n].copy()
    if kind == 5:       # repeats near the window limit
        blk = rng.integers(0, 256, int(rng.integers(30000, 32768)), dtype=np.uint8)
        return np.tile(blk, n // len(blk) + 1)[:n].copy()
    if kind == 6:       # text-like: Zipf over 60 symbols with word structure
        words = [rng.integers(97, 123, int(rng.integers(1, 10)), dtype=np.uint8) for _ in range(200)]
        idx = np.minimum(rng.zipf(1.3, n // 3 + 4) - 1, 199)
        return np.concatenate([np.append(words[i], 32) for i in idx])[:n].astype(np.uint8).copy()
    if kind == 7:       # 2-D patches in rows of a random width
        w = int(rng.integers(16, 2048))
        h = n // w + 1
        s = int(rng.integers(2, 64))
        small = rng.integers(0, 12, ((h + s - 1) // s, (w + s - 1) // s), dtype=np.uint8) * 10
        return np.kron(small, np.ones((s, s), np.uint8))[:h, :w].reshape(-1)[:n].copy()
    a = gen(rng, n)     # a mixture: noise injected into something else
    m = rng.random(n) < float(rng.uniform(0.001, 0.2))
    return np.where(m, rng.integers(0, 256, n, dtype=np.uint8), a).astype(np.uint8)


def compress(rng, raw):
    level = int(rng.integers(0, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
    wbits = int(rng.integers(9, 16))
    mem = int(rng.integers(1, 10))
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    out = []
    pos = 0
    b = raw.tobytes()
    while pos < len(b):
        step = int(rng.integers(1, max(2, len(b))))
        out.append(c.compress(b[pos:pos + step]))
        pos += step
        if rng.random() < 0.3:
            out.append(c.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH][int(rng.integers(0, 2))]))
    out.append(c.flush())
    return b"".join(out)


class _Args:
    pass


def run(seed=1, seconds=None, streams_budget=None, batch=256, verbose=True):
    """Runs until `streams_budget` random zlib streams are done (or `seconds` have passed).
    Returns (streams, decoded like zlib, refused, list of differences)."""
    a = _Args()
    a.seed, a.batch = seed, batch
    a.seconds = seconds if seconds is not None else 1e9
    budget = streams_budget if streams_budget is not None else 1 << 60
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    t_print = time.time()
    n_ok = n_refused = n_streams = 0
    bad = []
    with gpu.Engine(0) as e:
        while time.time() < t_end and n_streams < budget:
            streams, raws, rows = [], [], []
            # (rows of a power of two: tiles decoded in place; 4080: through their slots and untile_kernel)
            W = 4096 if rng.random() < 0.5 else 4080
            for k in range(a.batch):
                n = int(rng.integers(1, 300000)) if rng.random() < 0.9 else int(rng.integers(300000, 1 << 20))
                raw = gen(rng, n)
                st = compress(rng, raw)
                if rng.random() < 0.15:             # corrupt it
                    bst = bytearray(st)
                    for _ in range(int(rng.integers(1, 4))):
                        bst[int(rng.integers(0, len(bst)))] ^= 1 << int(rng.integers(0, 8))
                    st = bytes(bst)
                    d = zlib.decompressobj()
                    try:
                        ref = d.decompress(st, len(raw))
                    except zlib.error:
                        ref = None
                    raws.append(("corrupt", ref, len(raw)))
                else:
                    raws.append(("good", raw.tobytes(), len(raw)))
                streams.append(st)
                rows.append((len(raw) + W - 1) // W)
            tot_rows = sum(rows)
            wins, y = [], 0
            for k in range(a.batch):
                wins.append((0, 0, W, rows[k], 0, y))
                y += rows[k]
            # chunk of W x rows[k]; the last row is partly beyond the stream (reads as zeros)
            out, status = e.inflate_tiles(streams, W, rows, wins, (tot_rows, W))
            y = 0
            for k in range(a.batch):
                kind, ref, n = raws[k]
                got = out[y:y + rows[k]].reshape(-1)[:n].tobytes()
                y += rows[k]
                n_streams += 1
                if kind == "good":
                    if status[k] != 0 or got != ref:
                        bad.append(("good stream", k, int(status[k]), n))
                    else:
                        n_ok += 1
                else:
                    if status[k] != 0:
                        n_refused += 1
                    elif ref is not None and got[:len(ref)] != ref:
                        bad.append(("accepted corrupt stream decodes differently", k, 0, n))
                    else:
                        n_ok += 1
            if bad:
                break
            if verbose and time.time() - t_print > 60:
                t_print = time.time()
                print("... %d streams so far, no difference" % n_streams, flush=True)
    return n_streams, n_ok, n_refused, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=None)
    ap.add_argument("--streams", type=int, default=None, help="fixed budget of random streams (instead of a time)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    if a.seconds is None and a.streams is None:
        a.seconds = 60.0
    n_streams, n_ok, n_refused, bad = run(a.seed, a.seconds, a.streams, a.batch)
    print("seed %d: streams %d, decoded like zlib %d, refused %d, differences %d %s" % (a.seed, n_streams, n_ok, n_refused,
                                                                                       len(bad), bad[:5]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

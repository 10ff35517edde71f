"""GPU zlib decoding of landcover tiles (include/gcn10_gpu.h, gcn10_gpu_inflate_tiles).

The checker is Python's zlib (stock zlib): every stream it produces -- stored, fixed and
dynamic blocks, every level / strategy / window size, multi-block streams, matches at the
32 KiB limit, self-overlapping matches -- must decode on the GPU to the same bytes, through
the C ABI.  Malformed streams must end with a status, not with a fault.
"""
import zlib

import numpy as np
import pytest

from gcn10_amd import gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    with gpu.Engine(0) as e:
        yield e


def _data(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "classes":          # landcover-like: 11 class codes in runs
        codes = np.array([10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], np.uint8)
        runs = rng.geometric(0.05, size=n // 8 + 16)
        vals = codes[rng.integers(0, len(codes), len(runs))]
        return np.repeat(vals, runs)[:n].copy()
    if kind == "patches":          # 2-D patches, as a 1024-wide tile
        w = 1024
        h = (n + w - 1) // w
        small = rng.integers(0, 11, ((h + 31) // 32, w // 32), dtype=np.uint8) * 10
        return np.kron(small, np.ones((32, 32), np.uint8))[:h].reshape(-1)[:n].copy()
    if kind == "skewed":           # long codes: geometric symbol frequencies
        return np.minimum(rng.geometric(0.35, n), 255).astype(np.uint8)
    if kind == "period":           # self-overlapping matches of many periods
        out = []
        while sum(len(o) for o in out) < n:
            p = int(rng.integers(1, 70))
            out.append(np.tile(rng.integers(0, 256, p, dtype=np.uint8), int(rng.integers(2, 40))))
        return np.concatenate(out)[:n].copy()
    if kind == "far":              # repeats at the far end of the 32 KiB window
        blk = rng.integers(0, 256, 32768 - 300, dtype=np.uint8)
        return np.tile(blk, n // len(blk) + 1)[:n].copy()
    raise ValueError(kind)


def _compress(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    if not flush_every:
        return c.compress(raw.tobytes()) + c.flush()
    out = []
    for k in range(0, len(raw), flush_every):
        out.append(c.compress(raw[k:k + flush_every].tobytes()))
        out.append(c.flush(zlib.Z_FULL_FLUSH if (k // flush_every) % 2 else zlib.Z_SYNC_FLUSH))
    out.append(c.flush())
    return b"".join(out)


def _one(engine, stream, n, w=None):
    w = w or n
    rows = max(n // w, 1)
    out, status = engine.inflate_tiles([stream], w, [rows], [(0, 0, w, rows, 0, 0)], (rows, w))
    return out.reshape(-1), int(status[0])


@pytest.mark.parametrize("kind", ["zeros", "noise", "classes", "patches", "skewed", "period", "far"])
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_levels_and_kinds(engine, kind, level):
    n = 1024 * 256
    raw = _data(kind, n, seed=level)
    got, status = _one(engine, _compress(raw, level), n, 1024)
    assert status == 0
    assert np.array_equal(got, raw)


@pytest.mark.parametrize("strategy", [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
@pytest.mark.parametrize("kind", ["classes", "skewed", "noise"])
def test_strategies(engine, strategy, kind):
    n = 100000
    raw = _data(kind, n, seed=3)
    got, status = _one(engine, _compress(raw, 6, strategy), n)
    assert status == 0 and np.array_equal(got, raw)


@pytest.mark.parametrize("wbits", [9, 10, 12, 15])
def test_window_sizes(engine, wbits):
    raw = _data("period", 70000, seed=wbits)
    got, status = _one(engine, _compress(raw, 9, wbits=wbits), len(raw))
    assert status == 0 and np.array_equal(got, raw)


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 255, 258, 259, 4095, 16384, 16385, 32768, 32769, 65536, 1 << 20])
def test_sizes(engine, n):
    raw = _data("classes", n, seed=n)
    got, status = _one(engine, _compress(raw), n)
    assert status == 0 and np.array_equal(got, raw)


def test_multi_block_streams_with_flushes(engine):
    raw = np.concatenate([_data("classes", 50000, 1), _data("noise", 70000, 2), _data("zeros", 40000), _data("skewed", 60000, 3)])
    for every in (1000, 7777, 65536):
        got, status = _one(engine, _compress(raw, 6, flush_every=every), len(raw))
        assert status == 0 and np.array_equal(got, raw), every
    # stored blocks of every alignment: level 0 splits at 65535 bytes
    got, status = _one(engine, _compress(raw, 0, flush_every=333), len(raw))
    assert status == 0 and np.array_equal(got, raw)


def test_stream_shorter_and_longer_than_the_chunk(engine):
    raw = _data("classes", 5000, 5)
    st = _compress(raw)
    got, status = _one(engine, st, 8192)                 # short stream: zeros follow, as tiff.c does on the host
    assert status == 0 and np.array_equal(got[:5000], raw) and not got[5000:].any()
    out, status = engine.inflate_tiles([st], 100, [30], [(0, 0, 100, 30, 0, 0)], (30, 100))
    assert int(status[0]) == 0 and np.array_equal(out.reshape(-1), raw[:3000])     # long stream: cut


def test_many_tiles_with_windows(engine):
    rng = np.random.default_rng(11)
    tw, th = 256, 200
    nx, ny = 9, 7
    world = np.zeros((ny * th, nx * tw), np.uint8)
    streams, rows, wins = [], [], []
    # destination: a window of the world that cuts tiles on every side
    x0, y0, W, H = 100, 37, nx * tw - 333, ny * th - 111
    for ty in range(ny):
        for tx in range(nx):
            kind = ["classes", "noise", "zeros", "patches", "skewed"][(tx + ty) % 5]
            t = _data(kind, tw * th, seed=tx * 100 + ty).reshape(th, tw)
            world[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = t
            lvl = int(rng.integers(0, 10))
            streams.append(_compress(t.reshape(-1), lvl))
            rows.append(th)
            xs, xe = max(x0, tx * tw), min(x0 + W, (tx + 1) * tw)
            ys, ye = max(y0, ty * th), min(y0 + H, (ty + 1) * th)
            wins.append((xs - tx * tw, ys - ty * th, xe - xs, ye - ys, xs - x0, ys - y0))
    out, status = engine.inflate_tiles(streams, tw, rows, wins, (H, W))
    assert not status.any()
    assert np.array_equal(out, world[y0:y0 + H, x0:x0 + W])


def test_raw_and_predictor2_tiles_mixed_with_deflate_ones(engine):
    """Round 3 (ABI 3, gcn10_inflate_tile.flags): uncompressed chunks are untiled from the staged bytes --
    whole, or staged from the first wanted pixel on --, and chunks written with TIFF predictor 2 are summed
    back row by row, modulo 256, from each row's first pixel (tiff.c decode_chunk is the host form).  One
    launch holds all four kinds; chunk widths that are and are not multiples of the 1024-byte trip."""
    rng = np.random.default_rng(21)
    for tw, th in ((256, 200), (1024, 64), (1500, 9), (37, 50)):
        nx, ny = 5, 4
        world = np.zeros((ny * th, nx * tw), np.uint8)
        x0, y0, W, H = tw // 3, 7, nx * tw - tw // 2 - 11, ny * th - 13
        streams, rows, wins, flags, out_lens = [], [], [], [], []
        for ty in range(ny):
            for tx in range(nx):
                kind = ["classes", "noise", "patches", "skewed"][(tx + 2 * ty) % 4]
                t = _data(kind, tw * th, seed=tx * 10 + ty).reshape(th, tw)
                world[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = t
                xs, xe = max(x0, tx * tw), min(x0 + W, (tx + 1) * tw)
                ys, ye = max(y0, ty * th), min(y0 + H, (ty + 1) * th)
                sx, sy, cw, ch = xs - tx * tw, ys - ty * th, xe - xs, ye - ys
                diff = t.astype(np.int16)
                diff[:, 1:] -= t[:, :-1].astype(np.int16)
                diff = (diff & 0xFF).astype(np.uint8)
                mode = (tx + ty) % 4
                if mode == 0:                               # DEFLATE, no predictor (ABI 1 behaviour)
                    streams.append(_compress(t.reshape(-1), int(rng.integers(1, 10))))
                    flags.append(0)
                    out_lens.append(tw * th)
                elif mode == 1:                             # DEFLATE + predictor 2
                    streams.append(_compress(diff.reshape(-1), 6))
                    flags.append(gpu.TILE_PREDICTOR2)
                    out_lens.append(tw * th)
                elif mode == 2:                             # raw, the whole chunk staged
                    streams.append(t.tobytes())
                    flags.append(gpu.TILE_RAW)
                    out_lens.append(tw * th)
                else:                                       # raw, staged from the first wanted pixel to the last
                    flat = t.reshape(-1)
                    first, last = sy * tw + sx, (sy + ch - 1) * tw + sx + cw
                    streams.append(flat[first:last].tobytes())
                    flags.append(gpu.TILE_RAW)
                    out_lens.append(last - first)
                    sx, sy = 0, 0
                rows.append(th)
                wins.append((sx, sy, cw, ch, xs - x0, ys - y0))
        out, status = engine.inflate_tiles(streams, tw, rows, wins, (H, W), flags=flags, out_lens=out_lens)
        assert not status.any(), (tw, th, status)
        assert np.array_equal(out, world[y0:y0 + H, x0:x0 + W]), (tw, th)
    # a raw chunk shorter than its window is refused, not read past
    out, status = engine.inflate_tiles([bytes(100)], 20, [10], [(0, 0, 20, 10, 0, 0)], (10, 20), flags=[gpu.TILE_RAW],
                                       out_lens=[200])
    assert int(status[0]) == 8 and not out.any()


def test_full_size_tiles_of_a_block_row(engine):
    # 36 tiles of 1024 x 1024, the shape of one tile row of an ESA WorldCover file
    tiles = [_data(["patches", "classes"][k % 2], 1 << 20, seed=k) for k in range(36)]
    streams = [_compress(t, 6) for t in tiles]
    wins = [(0, 0, 1024, 1024, 1024 * k, 0) for k in range(36)]
    out, status = engine.inflate_tiles(streams, 1024, [1024] * 36, wins, (1024, 36 * 1024))
    assert not status.any()
    for k in range(36):
        assert np.array_equal(out[:, 1024 * k:1024 * (k + 1)].reshape(-1), tiles[k]), k


class _FixedWriter:
    """A zlib stream of ONE fixed-Huffman block (RFC 1951 3.2.6) from an explicit token list, so that a test
    decides every match's length and distance itself (zlib's own parser would not)."""
    _LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    _LEXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
    _DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
              6145, 8193, 12289, 16385, 24577]
    _DEXTRA = [0, 0, 0, 0] + [k for k in range(1, 14) for _ in (0, 1)]

    def __init__(self):
        self.acc, self.n, self.out, self.raw = 0, 0, bytearray(b"\x78\x01"), bytearray()
        self._bits(1, 1)        # BFINAL
        self._bits(1, 2)        # BTYPE = 01

    def _bits(self, v, n):          # LSB first
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def _code(self, code, n):       # Huffman codes go in MSB first
        self._bits(int(format(code, "0%db" % n)[::-1], 2), n)

    def _litlen(self, sym):
        if sym < 144:
            self._code(0x30 + sym, 8)
        elif sym < 256:
            self._code(0x190 + sym - 144, 9)
        elif sym < 280:
            self._code(sym - 256, 7)
        else:
            self._code(0xc0 + sym - 280, 8)

    def literal(self, b):
        self._litlen(b)
        self.raw.append(b)

    def match(self, length, dist):
        assert 3 <= length <= 258 and 1 <= dist <= len(self.raw) and dist <= 32768
        k = max(i for i, b in enumerate(self._LBASE) if b <= length) if length < 258 else 28
        self._litlen(257 + k)
        self._bits(length - self._LBASE[k], self._LEXTRA[k])
        d = max(i for i, b in enumerate(self._DBASE) if b <= dist)
        self._code(d, 5)
        self._bits(dist - self._DBASE[d], self._DEXTRA[d])
        for _ in range(length):
            self.raw.append(self.raw[-dist])

    def finish(self):
        self._litlen(256)
        if self.n:
            self._bits(0, 8 - self.n)
        return bytes(self.out) + zlib.adler32(bytes(self.raw)).to_bytes(4, "big"), bytes(self.raw)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_matches_of_every_length_from_far_and_near_sources(engine, seed):
    """Round 3: the copier carries out, all at once, every match of up to 64 bytes whose source lies before the
    batch at hand (dword steps + tail bytes, inside the 8 KiB ring), longer such matches one by one with every lane
    at work, and the others in order.  A hand-made token stream drives each kind through the ring many times: every
    length 3..258, distances from just behind the batch to the far end of the ring and beyond it (sources already
    flushed to HBM), sources that end exactly where the batch starts, matches that read the match before them,
    short-period repeats, single literals in between."""
    rng = np.random.default_rng(seed)
    w = _FixedWriter()
    for _ in range(9000):
        w.literal(int(rng.integers(0, 256)))
    lengths = list(range(3, 259)) * 3
    rng.shuffle(lengths)
    for k, length in enumerate(lengths):
        have = len(w.raw)
        kind = k % 8
        if kind == 0:
            dist = int(rng.integers(length, length + 200))      # near: may read tokens of the same batch
        elif kind == 1:
            dist = int(rng.integers(2500, 7000))                # before the batch, inside the ring
        elif kind == 2:
            dist = int(rng.integers(8100, min(have, 32768)))    # beyond the ring: read back from HBM
        elif kind == 3:
            dist = int(rng.integers(1, max(2, min(length, 64))))   # reads its own output
        elif kind == 4:
            dist = length                                       # source ends where the match starts
        elif kind == 5:
            dist = int(rng.integers(7600, 8200))                # around the ring's size
        elif kind == 6:
            dist = 1024                                         # the row above of a 1024-px tile
        else:
            dist = int(rng.integers(length, length + 2200))     # around the sub-batch limit
        w.match(length, min(dist, have))
        if k % 5 == 0:
            w.literal(int(rng.integers(0, 256)))
    while len(w.raw) % 1024:
        w.literal(int(rng.integers(0, 256)))
    stream, raw = w.finish()
    assert zlib.decompress(stream) == raw
    out, status = _one(engine, stream, len(raw))                # one row: through the tile's slot and untile_kernel
    assert status == 0
    assert out.tobytes() == raw
    out, status = _one(engine, stream, len(raw), w=1024)        # rows of 1024: decoded in place (matches that read
    assert status == 0                                          # flushed bytes read them from the raster's rows)
    assert out.tobytes() == raw


def _zlib_result(stream, n):
    d = zlib.decompressobj()
    try:
        return d.decompress(stream, n)
    except zlib.error:
        return None


def test_malformed_streams_report_a_status(engine):
    rng = np.random.default_rng(5)
    raw = _data("classes", 60000, 9)
    good = _compress(raw, 6)
    cases = [b"", b"\x78", b"\x78\x9c", b"\x00" * 64, b"\xff" * 64, b"\x78\x9c\x07" + b"\x00" * 10,
             good[: len(good) // 2], good[:10], b"\x78\x9c\x01\x05\x00\xfa\xfe" + b"abcde",      # bad NLEN
             bytes([0x78, 0x9C]) + bytes(rng.integers(0, 256, 500, dtype=np.uint8))]
    for k in range(40):                             # single-byte corruptions of a good stream
        b = bytearray(good)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(b))
    out_rows = len(raw) // 100
    streams = cases
    wins = [(0, 0, 100, out_rows, 0, k * out_rows) for k in range(len(streams))]
    out, status = engine.inflate_tiles(streams, 100, [out_rows] * len(streams), wins, (out_rows * len(streams), 100))
    assert status[:9].all(), status[:9]              # the hand-made ones are all invalid
    for k, st in enumerate(streams):
        if status[k] == 0:                          # accepted: then it decodes as zlib decodes it
            ref = _zlib_result(st, len(raw))
            got = out[k * out_rows:(k + 1) * out_rows].reshape(-1)
            if ref is not None:
                assert bytes(got[:len(ref)]) == ref, k
    # a distance that reaches before the start of the data
    # fixed block whose first symbol is a match (length 3, distance 1): nothing to copy from
    _, st = _one(engine, bytes([0x78, 0x9C, 0x03, 0x02, 0, 0, 0, 0]), 100)
    assert st == 6


def test_window_outside_its_chunk_is_refused(engine):
    raw = _data("classes", 100 * 30, 4)
    st = _compress(raw)
    for win in [(90, 0, 20, 5, 0, 0), (0, 28, 10, 5, 0, 0), (101, 0, 1, 1, 0, 0)]:
        out, status = engine.inflate_tiles([st], 100, [30], [win], (40, 40))
        assert int(status[0]) == 8 and not out.any(), win
    out, status = engine.inflate_tiles([st], 100, [30], [(90, 25, 10, 5, 3, 2)], (40, 40))
    assert int(status[0]) == 0 and np.array_equal(out[2:7, 3:13], raw.reshape(30, 100)[25:, 90:])


def test_argument_errors(engine):
    with pytest.raises(gpu.Gcn10GpuError):
        engine._chk(gpu.lib().gcn10_gpu_inflate_tiles(engine._ctx, None, None, 3, 100, None, 100, None, None), "x")
    assert gpu.lib().gcn10_gpu_inflate_tiles(engine._ctx, None, None, 0, 100, None, 100, None, None) == 0

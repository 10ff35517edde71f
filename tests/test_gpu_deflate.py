"""GPU tile DEFLATE (include/gcn10_gpu.h, gcn10_gpu_deflate_strip): every stream must inflate,
with stock zlib, to exactly the 256x256 tile (zero padded at raster edges) it encodes."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CN_VALUES = np.array([0, 15, 30, 35, 41, 48, 51, 55, 59, 62, 68, 72, 77, 83, 98, 255], dtype=np.uint8)


def _rasters(kind, H, W, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return np.full((H, W), 255, np.uint8)
    if kind == "zeros":
        return np.zeros((H, W), np.uint8)
    if kind == "patches":           # 25-px soil cells x landcover patches, like a CN raster
        a = rng.choice(CN_VALUES, size=((H + 24) // 25, (W + 24) // 25))
        return np.repeat(np.repeat(a, 25, axis=0), 25, axis=1)[:H, :W].copy()
    if kind == "noisy":             # i.i.d. over the CN value set: ~4 bits/px of entropy
        return rng.choice(CN_VALUES, size=(H, W)).astype(np.uint8)
    if kind == "random":            # incompressible: the stored-block fallback
        return rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    if kind == "rows":              # every row repeats the one above (distance-256 matches only)
        return np.repeat(rng.integers(0, 256, size=(1, W), dtype=np.uint8), H, axis=0)
    if kind == "skewed":            # one dominant symbol + rare ones: deep Huffman trees
        v = np.full((H, W), 77, np.uint8)
        idx = rng.random((H, W))
        for k, thr in enumerate([1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 3e-4, 1e-4, 5e-5, 2e-5]):
            v[idx < thr] = k
        return v
    if kind == "manyvals":          # > 144 live symbols but long runs: the flat-code path
        a = rng.integers(0, 256, size=((H + 7) // 8, (W + 7) // 8), dtype=np.uint8)
        return np.repeat(np.repeat(a, 8, axis=0), 8, axis=1)[:H, :W].copy()
    raise ValueError(kind)


def _check(engine, rasters, W, H):
    bufs = [engine.upload(r) for r in rasters]
    data, table, used = engine.deflate_rasters([b.ptr for b in bufs], W, H)
    for b in bufs:
        b.close()
    across, down = (W + 255) // 256, (H + 255) // 256
    assert table.shape == (len(rasters), down, across, 2)
    total = 0
    for r, img in enumerate(rasters):
        for ty in range(down):
            for tx in range(across):
                off, size = int(table[r, ty, tx, 0]), int(table[r, ty, tx, 1])
                assert off != 0xFFFFFFFF and 0 < size <= 65552 and off % 16 == 0 and off + size <= used
                want = np.zeros((256, 256), np.uint8)
                part = img[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                want[:part.shape[0], :part.shape[1]] = part
                got = zlib.decompress(data[off:off + size].tobytes())
                assert got == want.tobytes(), (r, ty, tx)
                total += size
    return total


@pytest.mark.parametrize("kind", ["uniform", "zeros", "patches", "noisy", "random", "rows", "skewed", "manyvals"])
def test_streams_inflate_to_the_tiles(engine, kind):
    H, W = 512, 768
    img = _rasters(kind, H, W, 1)
    size = _check(engine, [img], W, H)
    ref = sum(len(zlib.compress(img[y:y + 256, x:x + 256].tobytes(), 6))
              for y in range(0, H, 256) for x in range(0, W, 256))
    print("%-8s gpu %8d B   zlib-6 %8d B   ratio gpu/zlib %.2f   vs raw %.3f"
          % (kind, size, ref, size / ref, size / img.size))
    if kind in ("uniform", "zeros"):
        assert size < 6 * 300           # a constant tile: 256 one-token rows, ~250 B (266:1)
    if kind == "random":
        assert size == 6 * 65552        # stored fallback, never worse than raw + 16 B
    if kind in ("patches", "noisy", "rows", "skewed", "manyvals"):
        assert size < 2.0 * ref         # within 2x of zlib level 6 on CN-like data


@pytest.mark.parametrize("shape", [(1, 1), (256, 256), (255, 257), (300, 700), (513, 1025), (1000, 36001 // 16)])
def test_edge_tiles_and_odd_widths(engine, shape):
    H, W = shape
    imgs = [_rasters("patches", H, W, 3), _rasters("noisy", H, W, 4), _rasters("uniform", H, W, 5)]
    _check(engine, imgs, W, H)


def test_eighteen_rasters_in_one_launch(engine):
    H, W = 512, 1040
    imgs = [_rasters(["patches", "noisy", "skewed"][i % 3], H, W, 10 + i) for i in range(18)]
    _check(engine, imgs, W, H)


def test_argument_errors(engine):
    from gcn10_amd import gpu
    with pytest.raises(gpu.Gcn10GpuError):
        engine.deflate_rasters([], 256, 256)
    assert gpu.lib().gcn10_gpu_deflate_arena_bound(256, 256, 1) == 65552 + 4096      # + the pad that brings a raster's extent to a multiple of 4096 (round 3)
    assert gpu.lib().gcn10_gpu_deflate_arena_bound(257, 256, 2) == 4 * 65552 + 2 * 4096
    assert gpu.lib().gcn10_gpu_deflate_arena_bound(0, 256, 1) == 0


@pytest.mark.parametrize("kind", ["patches", "noisy", "skewed", "manyvals", "rows", "uniform"])
def test_wave_and_thread_code_construction_agree(engine, kind):
    """Pass B exists in two forms (one wave per tile, one thread per tile) that implement the same
    algorithm with the same tie breaks: every tile's stream must be byte-identical."""
    H, W = 512, 768
    img = _rasters(kind, H, W, 7)
    buf = engine.upload(img)
    streams = {}
    for mode in (1, 0):
        engine.set_option("deflate_wave_codes", mode)
        data, table, used = engine.deflate_rasters([buf.ptr], W, H)
        streams[mode] = [data[int(o):int(o) + int(n)].tobytes() for o, n in table.reshape(-1, 2)]
    engine.set_option("deflate_wave_codes", 1)
    buf.close()
    assert streams[0] == streams[1]


# ---- fused tile encoder: landcover + soil -> zlib streams, no CN raster in between -----------

def _fused_case(engine, tables, H, W, seed, cond_mask=3, table_mask=0x1FF, nasty=True, coherent=True,
                esa_override=None):
    from gcn10_amd import host
    from oracle import cn_oracle_c as oc
    from tests.util import make_block
    esa, gt, coarse, sgt = make_block(seed, H, W, H // 25 + 2, W // 25 + 2, nasty=nasty)
    if coherent:        # patchy landcover so that matches exist
        small = esa[::8, ::8]
        esa = np.ascontiguousarray(np.repeat(np.repeat(small, 8, axis=0), 8, axis=1)[:H, :W])
    if esa_override is not None:
        esa = esa_override
    hsy, hsx = coarse.shape
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
    data, table, used = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr, cond_mask, table_mask)
    for b in bufs:
        b.close()
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=cond_mask, table_mask=table_mask)
    sel = [r for r in range(18) if (cond_mask >> (r // 9)) & 1 and (table_mask >> (r % 9)) & 1]
    across, down = (W + 255) // 256, (H + 255) // 256
    assert table.shape == (len(sel), down, across, 2)
    total = 0
    for j, r in enumerate(sel):
        for ty in range(down):
            for tx in range(across):
                off, size = int(table[j, ty, tx, 0]), int(table[j, ty, tx, 1])
                assert off != 0xFFFFFFFF and 0 < size <= 65552 and off + size <= used
                exp = np.zeros((256, 256), np.uint8)
                part = want[r][ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                exp[:part.shape[0], :part.shape[1]] = part
                assert zlib.decompress(data[off:off + size].tobytes()) == exp.tobytes(), (r, ty, tx)
                total += size
    return total, want, sel


@pytest.mark.parametrize("shape", [(256, 256), (300, 700), (513, 1025), (1, 1), (700, 36001 // 40)])
def test_fused_encoder_streams_inflate_to_the_oracle_rasters(engine, tables, shape):
    H, W = shape
    _fused_case(engine, tables, H, W, seed=H + W)


def test_fused_encoder_noise_and_subsets(engine, tables):
    # i.i.d. landcover: mostly literals, big streams (the 64 KB-image variant)
    _fused_case(engine, tables, 300, 520, seed=5, coherent=False)
    _fused_case(engine, tables, 300, 520, seed=6, cond_mask=2, table_mask=0x0A1)
    _fused_case(engine, tables, 260, 260, seed=7, cond_mask=1, table_mask=0x100)


def test_fused_encoder_awkward_tables(engine):
    from tests.util import random_tables
    t = random_tables(77, 9)
    t[:, 6:, :] = 255           # 6 live classes x 36 soil-code pairs: under 256 distinct 18-vectors
    from gcn10_amd import host
    H, W = 300, 300
    rng = np.random.default_rng(8)
    # landcover from the live classes (and a few dead ones), patchy
    small = rng.integers(0, 9, size=((H + 7) // 8, (W + 7) // 8), dtype=np.uint8)
    _fused_case(engine, t, H, W, seed=8, esa_override=np.ascontiguousarray(
        np.repeat(np.repeat(small, 8, axis=0), 8, axis=1)[:H, :W]))


@pytest.mark.parametrize("pattern", ["natural", "iid"])
def test_fused_encoder_full_block_width(engine, tables, pattern):
    """Two tile rows of a 36000-px wide block (141 tile positions across, the last one 160 px
    wide), landcover of the bench patterns: all 5076 streams inflate to the oracle's rasters."""
    import bench
    esa, _, _, _ = bench.synth_block(3, 4096, pattern)
    esa = np.ascontiguousarray(np.tile(esa[:512], (1, 9))[:, :36000])
    _fused_case(engine, tables, 512, 36000, seed=21, nasty=False, coherent=False, esa_override=esa)


@pytest.mark.parametrize("texture", ["natural", "patches", "iid", "runs", "rows", "constant", "stripes63"])
def test_segment_parallel_parse_is_the_row_parse_bit_for_bit(engine, tables, texture):
    """Round 3: pass F-A with one lane per 64-pixel segment (speculative parse + entry offsets, option
    fused_parse=1, the default) writes the same token stream as one lane per row (fused_parse=0), and pass
    F-C with every wave packing its own quarter of the tokens (fused_emit=1, the default) the same bits as
    the lock-step form: arena bytes, stream table and bytes used are identical in all four combinations, for
    textures whose matches end inside, at and far beyond segment boundaries (whole-row matches, runs of
    63 / 64 / 65 pixels, vertical repeats)."""
    import bench
    from gcn10_amd import host
    H, W = 512, 2 * 256 + 100
    rng = np.random.default_rng(31)
    classes = np.array([10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], np.uint8)
    if texture in ("natural", "patches", "iid"):
        esa = np.ascontiguousarray(bench.synth_block(4, 1024, texture)[0][:H, :W])
    elif texture == "runs":          # runs of 1 .. 200 pixels: matches cross one, two and three boundaries
        runs = rng.integers(1, 200, size=H * W // 20)
        esa = np.repeat(classes[rng.integers(0, len(classes), len(runs))], runs)[:H * W].reshape(H, W).copy()
    elif texture == "rows":          # every row a copy of the one above, except where it is not
        esa = np.tile(classes[rng.integers(0, len(classes), W)], (H, 1))
        esa[rng.integers(0, H, 40), rng.integers(0, W, 40)] = 0
        esa = np.ascontiguousarray(esa)
    elif texture == "constant":
        esa = np.full((H, W), 80, np.uint8)
        esa[100, 300] = 10
    else:                            # stripes of 63, 64 and 65 pixels: matches that end just before, at and after a boundary
        row = np.concatenate([np.full(n, classes[k % len(classes)], np.uint8) for k, n in enumerate([63, 64, 65, 1, 2, 64, 127, 128, 98])])
        esa = np.ascontiguousarray(np.tile(row[:W], (H, 1)))
        esa[::7, ::61] = 30
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], np.uint8), size=(H // 25 + 2, W // 25 + 2))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [-0.01, 3.02 / coarse.shape[1], 0.0, 3.01, 0.0, -3.02 / coarse.shape[0]]
    ci, cj = host.build_index_maps(gt, sgt, W, H, coarse.shape[1], coarse.shape[0])
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, coarse.shape[1], coarse.shape[0], bufs[2].ptr, W)
    out = {}
    try:
        # (pass F-A form, pass F-C form): rounds 1-2 = (0, 0); round 3's defaults = (1, 1)
        for parse, emit in ((0, 0), (1, 0), (0, 1), (1, 1)):
            engine.set_option("fused_parse", parse)
            engine.set_option("fused_emit", emit)
            out[parse, emit] = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr)
    finally:
        engine.set_option("defaults", 0)
        for b in bufs:
            b.close()
    ref = out[0, 0]
    for key in ((1, 0), (0, 1), (1, 1)):
        assert ref[2] == out[key][2], "%s: bytes used differ: %d vs %d" % (key, ref[2], out[key][2])
        assert np.array_equal(ref[1], out[key][1]), key
        assert np.array_equal(ref[0], out[key][0]), key


@pytest.mark.parametrize("seg_align", [4096, 16])
def test_streams_placed_by_the_bit_packing_pass_lie_where_the_placement_pass_puts_them(engine, tables, seg_align):
    """Round 3: the wave-independent pass F-C places its streams itself (chunk totals from pass B, no pass B'
    launch).  A strip of 992 tile positions (16 chunks of 64: more chunk totals than a workgroup has lanes),
    ragged right edge, half of it without dual soil classes (aliases whose original belongs to another
    group's workgroup): table, arena bytes and bytes used equal those of the lock-step form behind pass B'."""
    import bench
    from gcn10_amd import host
    H, W = 31 * 256, 31 * 256 + 17
    rng = np.random.default_rng(77)
    esa = np.ascontiguousarray(np.tile(bench.synth_block(5, 2048, "patches")[0], (4, 4))[:H, :W])
    hsy, hsx = H // 25 + 2, W // 25 + 2
    coarse = rng.choice(np.array([1, 2, 3, 4], np.uint8), size=(hsy, hsx))
    coarse[:, hsx // 2:] = rng.choice(np.array([0, 1, 2, 11, 12, 13, 14, 255], np.uint8), size=(hsy, hsx - hsx // 2))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [-0.01, 3.02 / hsx, 0.0, 3.01, 0.0, -3.02 / hsy]
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
    out = {}
    try:
        for emit in (0, 1):
            engine.set_option("arena_segment_align", seg_align)
            engine.set_option("fused_emit", emit)
            out[emit] = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr)
    finally:
        engine.set_option("defaults", 0)
        for b in bufs:
            b.close()
    assert out[0][2] == out[1][2], "bytes used differ: %d vs %d" % (out[0][2], out[1][2])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])
    table = out[1][1].reshape(18, -1, 2).astype(np.int64)
    assert len({tuple(e) for e in table.reshape(-1, 2).tolist()}) < table.shape[0] * table.shape[1], "no aliases in this case?"
    # every raster's own streams: one contiguous extent in tile order that starts at a multiple of the alignment
    end_before = 0
    for r in range(18):
        alias = np.zeros(table.shape[1], bool)        # an alias carries an earlier raster's entry of the same position
        for q in range(r):
            alias |= table[q][:, 0] == table[r][:, 0]
        own = table[r][~alias]
        if len(own) == 0:
            continue
        assert own[0, 0] % seg_align == 0 and own[0, 0] >= end_before
        slots = (own[:, 1] + 15) // 16 * 16
        assert np.array_equal(own[1:, 0], own[0, 0] + np.cumsum(slots)[:-1])
        end_before = own[-1, 0] + slots[-1]
    assert end_before == out[1][2]


@pytest.mark.parametrize("emit", [0, 1])
def test_an_arena_too_small_for_the_strip_says_so_and_keeps_what_fits(engine, tables, emit):
    """include/gcn10_gpu.h: a stream that does not fit the caller's arena gets offset 0xffffffff and size 0, the
    bytes used are reported in full (the host sees used > capacity and gives the strip up), and nothing is written
    behind the arena's end.  Both placements: pass B' and the one inside pass F-C."""
    import bench
    from gcn10_amd import host
    H, W = 512, 1024 + 60
    rng = np.random.default_rng(5)
    esa = np.ascontiguousarray(bench.synth_block(6, 2048, "natural")[0][:H, :W])
    hsy, hsx = H // 25 + 2, W // 25 + 2
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], np.uint8), size=(hsy, hsx))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [0.0, 3.0 / hsx, 0.0, 3.0, 0.0, -3.0 / hsy]
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
    try:
        engine.set_option("fused_emit", emit)
        full_data, full_tab, full_used = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr)
        cap = (full_used * 3 // 5) // 4096 * 4096
        data, tab, used = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr, arena_cap=cap)
    finally:
        engine.set_option("defaults", 0)
        for b in bufs:
            b.close()
    assert used == full_used and used > cap
    full_tab, tab = full_tab.reshape(-1, 2), tab.reshape(-1, 2)
    fits = full_tab[:, 0].astype(np.int64) + (full_tab[:, 1].astype(np.int64) + 15) // 16 * 16 <= cap
    assert fits.any() and (~fits).any()
    assert np.array_equal(tab[fits], full_tab[fits])
    assert (tab[~fits, 0] == 0xffffffff).all() and (tab[~fits, 1] == 0).all()
    for off, size in tab[fits]:
        assert np.array_equal(data[off:off + size], full_data[off:off + size])


def test_fused_encoder_shares_streams_where_drained_equals_undrained(engine, tables):
    """No dual soil class (11..14) under a tile: the drained and the undrained raster of a table
    are the same bytes there, and the fused encoder emits them once -- both table entries point
    at one stream.  Tiles with a dual class keep two streams."""
    from gcn10_amd import host
    from oracle import cn_oracle_c as oc
    H, W = 512, 768
    rng = np.random.default_rng(14)
    small = rng.choice(np.array([10, 20, 30, 40, 50, 60, 80, 90], np.uint8), size=(H // 16, W // 16))
    esa = np.ascontiguousarray(np.repeat(np.repeat(small, 16, axis=0), 16, axis=1))
    hsy, hsx = H // 25 + 2, W // 25 + 2
    coarse = rng.choice(np.array([1, 2, 3, 4], np.uint8), size=(hsy, hsx))
    coarse[:, hsx // 2:] = rng.choice(np.array([1, 2, 11, 12, 13, 14], np.uint8), size=(hsy, hsx - hsx // 2))
    gt = [0.0, 0.001, 0.0, 1.0, 0.0, -0.001]
    sgt = [0.0, 0.025, 0.0, 1.0, 0.0, -0.025]
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
    data, table, used = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr)
    for b in bufs:
        b.close()
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    shared = 0
    for r in range(18):
        for ty in range(2):
            for tx in range(3):
                off, size = int(table[r, ty, tx, 0]), int(table[r, ty, tx, 1])
                assert zlib.decompress(data[off:off + size].tobytes()) == \
                    want[r][ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256].tobytes(), (r, ty, tx)
                if r >= 9 and off == int(table[r - 9, ty, tx, 0]):
                    shared += 1
    # the left tile column has no dual class anywhere near it: 9 tables x 2 tile rows share
    assert shared >= 18, shared
    assert all(np.array_equal(want[r][:, :256], want[r - 9][:, :256]) for r in range(9, 18))
    # where the two rasters differ nothing may be shared
    for r in range(9, 18):
        for ty in range(2):
            for tx in range(3):
                a = want[r][ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                b = want[r - 9][ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]
                if not np.array_equal(a, b):
                    assert int(table[r, ty, tx, 0]) != int(table[r - 9, ty, tx, 0])


def test_fused_encoder_emits_one_stream_where_all_rasters_agree(engine, tables):
    """Open water (class 80: CN 100 in every table), snow and ice (70: 0), no-data (0: 255): on a
    tile that holds nothing else all 18 rasters are the same bytes -- one stream, 18 table entries."""
    from gcn10_amd import host
    from oracle import cn_oracle_c as oc
    H, W = 256, 1024
    esa = np.zeros((H, W), np.uint8)
    esa[:, :256] = 80
    esa[:, 256:512] = 70
    esa[:128, 512:768] = 80                              # water and no-data: still one stream
    esa[:, 768:] = np.where(np.arange(256)[None, :] < 128, 80, 40)     # water and cropland: rasters differ
    coarse = np.full((12, 42), 2, np.uint8)              # soil group B everywhere
    gt = [0.0, 0.001, 0.0, 1.0, 0.0, -0.001]
    sgt = [0.0, 0.025, 0.0, 1.0, 0.0, -0.025]
    ci, cj = host.build_index_maps(gt, sgt, W, H, 42, 12)
    engine.set_tables(tables)
    bufs = [engine.upload(a) for a in (esa, coarse, ci, cj)]
    engine.prepare_tile(bufs[1].ptr, 42, 12, bufs[2].ptr, W)
    data, table, used = engine.deflate_fused(bufs[0].ptr, W, H, bufs[3].ptr)
    for b in bufs:
        b.close()
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    for r in range(18):
        for tx in range(4):
            off, size = int(table[r, 0, tx, 0]), int(table[r, 0, tx, 1])
            assert zlib.decompress(data[off:off + size].tobytes()) == want[r][:, tx * 256:(tx + 1) * 256].tobytes()
    for tx in range(3):
        assert len({int(table[r, 0, tx, 0]) for r in range(18)}) == 1, tx
    assert len({int(table[r, 0, 3, 0]) for r in range(18)}) > 1


def test_fused_vs_unfused_size(engine, tables):
    """The class-based match structure costs little compression against per-raster parsing."""
    H, W = 512, 768
    fused, want, sel = _fused_case(engine, tables, H, W, seed=9)
    bufs = [engine.upload(want[r]) for r in sel]
    data, table, used = engine.deflate_rasters([b.ptr for b in bufs], W, H)
    for b in bufs:
        b.close()
    unfused = int(table[..., 1].sum())
    print("fused %d B, per-raster %d B, ratio %.3f" % (fused, unfused, fused / unfused))
    assert fused <= 1.10 * unfused


def test_fused_needs_at_most_256_classes(engine):
    from gcn10_amd import gpu
    from tests.util import random_tables
    engine.set_tables(random_tables(3, 9))              # every (class, soil) pair its own vector
    esa = engine.upload(np.zeros((16, 16), np.uint8))
    coarse = engine.upload(np.zeros((1, 1), np.uint8))
    ci = engine.upload(np.zeros(16, np.int32))
    engine.prepare_tile(coarse.ptr, 1, 1, ci.ptr, 16)
    with pytest.raises(gpu.Gcn10GpuError, match="256 pixel classes"):
        engine.deflate_fused(esa.ptr, 16, 16, ci.ptr)
    for b in (esa, coarse, ci):
        b.close()

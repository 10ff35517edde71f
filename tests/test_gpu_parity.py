"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Bit-exact is the bar everywhere: this is uint8 / integer table work, and the
only floating point (fp64 index maps) is computed on the host by
libgcn10_host.so and checked against the oracle in tests/test_host.py.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from gcn10_amd import gpu, host
from oracle import cn_oracle_c as oc
from tests.conftest import GOLDEN
from tests.util import make_block, random_tables

pytestmark = pytest.mark.gpu


def same(got, want, what):
    """np.array_equal with a report of where two rasters differ (for the assertion message)."""
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    if got.shape == want.shape and np.array_equal(got, want):
        return True
    if got.shape != want.shape:
        print("MISMATCH %s: shapes %s / %s" % (what, got.shape, want.shape))
        return False
    d = np.flatnonzero(got != want)
    runs = np.split(d, np.flatnonzero(np.diff(d) > 1) + 1)
    print("MISMATCH %s: %d of %d bytes differ in %d run(s); first runs (start, length): %s; got %s want %s"
          % (what, d.size, got.size, len(runs), [(int(r[0]), int(r.size)) for r in runs[:6]],
             got[d[:12]].tolist(), want[d[:12]].tolist()))
    return False


@pytest.fixture
def eng9(engine, tables):
    """The shared engine with the nine shipped tables loaded (other tests load their own)."""
    engine.set_tables(tables)
    return engine


def test_device_is_gfx950(engine):
    info = engine.device_info()
    assert "gfx950" in info["name"] and info["cus"] >= 200


# ---- per-function kernels -------------------------------------------------

@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (17, 1), (1, 4097), (257, 131), (300, 300), (64, 1040)])
def test_resample_equals_oracle(engine, shape):
    H, W = shape
    esa, gt, coarse, sgt = make_block(H * 1000 + W, H, W, H // 25 + 2, W // 25 + 3, nasty=True)
    hsy, hsx = coarse.shape
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    c_d, ci_d, cj_d = engine.upload(coarse), engine.upload(ci), engine.upload(cj)
    out_d = engine.alloc(H * W)
    engine.memset(out_d.ptr, 0xA5, H * W)
    engine.resample(c_d.ptr, hsx, hsy, ci_d.ptr, cj_d.ptr, W, H, out_d.ptr)
    got = engine.download(out_d.ptr, (H, W))
    assert np.array_equal(got, oc.resample(coarse, gt, sgt, W, H))
    for b in (c_d, ci_d, cj_d, out_d):
        b.close()


@pytest.mark.parametrize("n,offset", [(1, 0), (15, 0), (16, 0), (4097, 0), (100003, 0), (100003, 3), (50, 13), (7, 9)])
@pytest.mark.parametrize("drained", [True, False])
def test_modify_hysogs_equals_oracle(engine, n, offset, drained):
    rng = np.random.default_rng(n + offset)
    h = rng.integers(0, 256, size=n, dtype=np.uint8)
    h[: min(n, 256)] = np.arange(min(n, 256), dtype=np.uint8)
    buf = engine.alloc(n + offset + 32)
    engine.memset(buf.ptr, 0x11, n + offset + 32)
    engine.h2d(buf.at(offset), h)
    engine.modify_hysogs_data(buf.at(offset), n, drained)
    allb = engine.download(buf.ptr, (n + offset + 32,))
    assert np.array_equal(allb[offset:offset + n], oc.modify_hysogs_data(h, drained))
    # neighbours untouched (0x11 = 17 is not a dual class, so a stray remap would not show;
    # check with the guard value instead)
    assert (allb[:offset] == 0x11).all() and (allb[offset + n:] == 0x11).all()
    buf.close()


@pytest.mark.parametrize("n,offset", [(1, 0), (16, 0), (31, 0), (65536 + 5, 0), (1000, 1), (1000, 16)])
def test_calculate_cn_equals_oracle(engine, n, offset):
    t = random_tables(n, 9)
    engine.set_tables(t)
    rng = np.random.default_rng(n)
    esa = rng.integers(0, 256, size=n, dtype=np.uint8)
    hsg = rng.choice(np.array([0, 1, 2, 3, 4, 5, 11, 14, 255], dtype=np.uint8), size=n)
    e_d, h_d, o_d = engine.alloc(n + 64), engine.alloc(n + 64), engine.alloc(n + 64)
    engine.h2d(e_d.at(offset), esa)
    engine.h2d(h_d.at(offset), hsg)
    for k in (0, 4, 8):
        engine.memset(o_d.ptr, 0xA5, n + 64)
        engine.calculate_cn(e_d.at(offset), h_d.at(offset), n, k, o_d.at(offset))
        got = engine.download(o_d.ptr, (n + 64,))
        assert np.array_equal(got[offset:offset + n], oc.calculate_cn(esa, hsg, t[k]))
        assert (got[:offset] == 0xA5).all() and (got[offset + n:] == 0xA5).all()
    for b in (e_d, h_d, o_d):
        b.close()


def test_all_65536_pairs_both_kernels(eng9, tables):
    """esa[y][x] = y, soil = x // 8: every (class, soil code) pair through all 18 rasters
    (2048 px wide: the 16-byte-per-lane kernels need rows of at least 1040 px)."""
    esa = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 2048, axis=1)
    coarse = np.arange(256, dtype=np.uint8)[None, :]
    gt = [0.0, 1.0, 0.0, 0.0, 0.0, -1.0]
    sgt = [4.0, 8.0, 0.0, 0.0, 0.0, -1.0]            # ci[x] = round((x + 0.5 - 4) / 8) = x // 8
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    assert all(len(np.unique(want[r])) > 1 for r in range(18))
    got = eng9.process_block_mem(esa, gt, coarse, sgt)
    assert eng9.last_kernel_name().startswith("cn_strip_kernel<0,")
    assert np.array_equal(got, want)
    # single-table kernel, every table
    for k in range(9):
        g1 = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=3, table_mask=1 << k)
        assert eng9.last_kernel_name().startswith("cn_strip_kernel<1,")
        assert np.array_equal(g1[k], want[k]) and np.array_equal(g1[9 + k], want[9 + k])


# ---- fused block path -----------------------------------------------------

SHAPES = [(1, 1), (1, 15), (1, 16), (1, 17), (17, 1), (5, 3), (3, 5000), (64, 64), (257, 131),
          (300, 300), (64, 1040), (33, 4099), (130, 1024), (9, 36001 // 9),
          (70, 36001),        # the real block width: rows not 16-byte aligned
          # widths around the vector kernel's lower bound (1040) and odd widths above it: lanes
          # whose 16 pixels straddle a row end, row ends inside a wave's 1024-px span, npix % 16 != 0
          (3, 1039), (5, 1041), (41, 1055), (7, 2047), (21, 2049), (2, 36001), (1, 1040), (1, 70001),
          (19, 8193)]


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_block_equals_oracle(eng9, tables, shape):
    H, W = shape
    esa, gt, coarse, sgt = make_block(H * 7919 + W, H, W, H // 25 + 2, W // 25 + 2, nasty=True)
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    got = eng9.process_block_mem(esa, gt, coarse, sgt)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("strip_rows", [1, 7, 16, 48, 1000])
def test_strips_equal_whole_block(eng9, tables, strip_rows):
    # W = 1041 is odd: strips that do not start on a 16-byte boundary take the byte kernel
    for W in (1040, 1041):
        esa, gt, coarse, sgt = make_block(W + strip_rows, 97, W, 6, 43, nasty=True)
        want = oc.process_block_mem(esa, gt, coarse, sgt, tables)
        got = eng9.process_block_mem(esa, gt, coarse, sgt, strip_rows=strip_rows)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("cond_mask", [1, 2, 3])
@pytest.mark.parametrize("table_mask", [0x1FF, 0x001, 0x080, 0x100, 0x0A5, 0x1FE])
def test_subsets(eng9, tables, cond_mask, table_mask):
    esa, gt, coarse, sgt = make_block(cond_mask * 1000 + table_mask, 70, 523, 5, 23, nasty=True)
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=cond_mask,
                                table_mask=table_mask)
    got = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=cond_mask, table_mask=table_mask)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cond_mask", [1, 2, 3])
@pytest.mark.parametrize("table_mask", [0x1FF, 0x001, 0x080, 0x0A5])
@pytest.mark.parametrize("W", [1043, 2064])
def test_subsets_vector_kernels(eng9, tables, cond_mask, table_mask, W):
    """The same subsets at widths the 16-byte-per-lane kernels take (1043: rows not 16-byte aligned)."""
    esa, gt, coarse, sgt = make_block(cond_mask * 1000 + table_mask + W, 37, W, 4, W // 25 + 2, nasty=True)
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=cond_mask, table_mask=table_mask)
    got = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=cond_mask, table_mask=table_mask)
    assert eng9.last_kernel_name().startswith("cn_strip_kernel<")
    assert np.array_equal(got, want)


@pytest.mark.parametrize("W,H,hsx,hsy,state", [
    (1040, 40, 50, 5, 1),          # narrowest width of the vector kernels; ~21 columns per soil cell
    (2048, 33, 82, 7, 1),
    (4112, 21, 170, 3, 1),
    (36000, 9, 1440, 2, 1),        # the BASELINE ratio (25 columns per cell)
    (2048, 30, 128, 4, 1),         # exactly 16 columns per cell (jittered grid: groups straddle two cells)
    (2064, 30, 200, 4, 2),         # ~10 columns per cell: some group spans three cells -> bytes for the tile
    (2048, 30, 1000, 4, 2),        # cells narrower than a pixel pair
    (2051, 30, 82, 4, 1),          # W % 16 != 0: words are written, the strips read bytes all the same
])
def test_compact_soil_words_change_no_raster(eng9, tables, W, H, hsx, hsy, state):
    """Strips of 16-byte aligned rows read one compact word per 16-px column group instead of 16 code bytes
    when every group of the tile has at most two soil cells; otherwise, and with the option off, the bytes."""
    esa, gt, coarse, sgt = make_block(W * 3 + hsx, H, W, hsy, hsx, nasty=True)
    cases = [(1, 1 << 1), (2, 1 << 7), (3, 1 << 4), (3, 0x1ff), (2, 0b100100101), (1, 0x1ff)]
    want = {c: oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=c[0], table_mask=c[1]) for c in cases}
    try:
        for on in (1, 0, 1):
            eng9.set_option("compact_soil", on)
            for c in cases:
                got = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=c[0], table_mask=c[1])
                assert eng9.soil_words_state() == (state if on else 0)
                assert np.array_equal(got, want[c]), (on, c)
        # strips (the second one starts inside a soil row) and the launch shapes
        eng9.set_option("compact_soil", 1)
        for ilp1, pf, bpc, xcd in ((1, 1, 2, 1), (2, 0, 16, 0), (4, 0, 8, 1), (2, 1, 8, 1)):
            for name, v in (("ilp1", ilp1), ("ilp16", min(ilp1, 2)), ("prefetch", pf), ("grid_blocks_per_cu", bpc),
                            ("xcd_slabs", xcd)):
                eng9.set_option(name, v)
            for c in cases[:4]:
                got = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=c[0], table_mask=c[1], strip_rows=13)
                assert np.array_equal(got, want[c]), (ilp1, pf, bpc, xcd, c)
    finally:
        eng9.set_option("defaults", 0)


def test_compact_soil_words_with_a_column_map_that_is_not_monotone(eng9, tables):
    """ci is the caller's: a decreasing map (a soil grid stored east to west) has compact groups as well, a
    zig-zag one has none; both must give coarse[cj[y]][ci[x]] exactly."""
    W, H, hsx, hsy = 2048, 24, 90, 6
    rng = np.random.default_rng(17)
    esa = rng.choice(np.array([10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100, 0, 255], dtype=np.uint8), size=(H, W))
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255, 7], dtype=np.uint8), size=(hsy, hsx))
    cj = np.minimum(np.arange(H) * hsy // H, hsy - 1).astype(np.int32)
    maps = {"decreasing": (hsx - 1 - np.arange(W) * hsx // W).astype(np.int32),
            "zig-zag": ((np.arange(W) // 5) % 2 * 40 + np.arange(W) * 40 // W).astype(np.int32),
            "constant": np.full(W, 17, dtype=np.int32)}
    e = eng9
    bufs = [e.upload(esa), e.upload(coarse), e.upload(cj)]
    outs = [e.alloc(W * H) for _ in range(2)]
    try:
        for name, ci in maps.items():
            d_ci = e.upload(ci)
            e.prepare_tile(bufs[1].ptr, hsx, hsy, d_ci.ptr, W)
            assert e.soil_words_state() == (2 if name == "zig-zag" else 1), name
            ptrs = [None] * 18
            ptrs[3], ptrs[9 + 3] = outs[0].ptr, outs[1].ptr
            e.cn_strip(bufs[0].ptr, W, H, bufs[2].ptr, 3, 1 << 3, ptrs)
            e.sync()
            soil = coarse[cj][:, ci]
            for c, drained in ((0, True), (1, False)):
                want = oc.calculate_cn(esa, oc.modify_hysogs_data(soil, drained), tables[3])
                assert np.array_equal(e.download(outs[c].ptr, (H, W)), want.reshape(H, W)), (name, c)
            d_ci.close()
    finally:
        for b in bufs + outs:
            b.close()


def test_launch_shape_knobs_never_change_results(eng9, tables):
    """gcn10_gpu_set_option: chunks per trip, software pipeline, store policy, XCD slabs, grid size."""
    W, H = 3001, 53
    esa, gt, coarse, sgt = make_block(4242, H, W, 5, 123, nasty=True)
    want18 = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    want1 = oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=2, table_mask=1 << 4)
    try:
        for ilp1, ilp16, pf, nt, xcd, bpc in [(1, 1, 0, 1, 1, 8), (1, 1, 1, 0, 0, 1), (2, 2, 1, 1, 1, 16),
                                              (2, 2, 0, 0, 1, 3), (4, 0, 1, 1, 0, 8), (4, 0, 0, 1, 1, 2)]:
            for name, v in (("ilp1", ilp1), ("ilp16", ilp16), ("prefetch", pf), ("nontemporal", nt),
                            ("xcd_slabs", xcd), ("grid_blocks_per_cu", bpc)):
                eng9.set_option(name, v)
            assert np.array_equal(eng9.process_block_mem(esa, gt, coarse, sgt), want18)
            got1 = eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=2, table_mask=1 << 4)
            assert np.array_equal(got1, want1)
    finally:
        eng9.set_option("defaults", 0)


def test_tune_single_raster_picks_a_position_and_changes_no_result(eng9, tables):
    W, H = 2051, 64
    esa, gt, coarse, sgt = make_block(99, H, W, 4, 90, nasty=True)
    want = oc.process_block_mem(esa, gt, coarse, sgt, tables, cond_mask=2, table_mask=1 << 3)
    hsy, hsx = coarse.shape
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    e = eng9
    bufs = [e.upload(esa), e.upload(coarse), e.upload(ci), e.upload(cj)]
    npix = W * H
    arena = e.alloc(npix + 5 * 4096)
    try:
        e.prepare_tile(bufs[1].ptr, hsx, hsy, bufs[2].ptr, W)
        best, ms, rep = e.tune_single_raster(bufs[0].ptr, W, H, bufs[3].ptr, 2, 1 << 3, arena.ptr, npix + 5 * 4096, 4096)
        assert arena.ptr <= best <= arena.ptr + 5 * 4096 and (best - arena.ptr) % 4096 == 0
        assert rep["positions"] == 6 and rep["best_ms"] == pytest.approx(ms, abs=1e-3) and 0 < ms <= rep["worst_ms"]
        ptrs = [None] * 18
        ptrs[9 + 3] = best
        e.cn_strip(bufs[0].ptr, W, H, bufs[3].ptr, 2, 1 << 3, ptrs)
        e.sync()
        assert np.array_equal(e.download(best, (H, W)), want[9 + 3])
        with pytest.raises(gpu.Gcn10GpuError):
            e.tune_single_raster(bufs[0].ptr, W, H, bufs[3].ptr, 3, 1 << 3, arena.ptr, npix, 4096)   # two conditions
        with pytest.raises(gpu.Gcn10GpuError):
            e.tune_single_raster(bufs[0].ptr, W, H, bufs[3].ptr, 1, 1 << 3, arena.ptr, npix - 1, 4096)   # arena too small
    finally:
        e.set_option("defaults", 0)
        for b in bufs + [arena]:
            b.close()


def test_stream_copy_is_a_copy(engine):
    n = 16 * 100003
    src = np.random.default_rng(5).integers(0, 256, size=n, dtype=np.uint8)
    a, b = engine.upload(src), engine.alloc(n + 32)
    engine.memset(b.ptr, 0x33, n + 32)
    engine.stream_copy(a.ptr, b.ptr, n)
    engine.sync()
    got = engine.download(b.ptr, (n + 32,))
    assert np.array_equal(got[:n], src) and (got[n:] == 0x33).all()
    with pytest.raises(gpu.Gcn10GpuError):
        engine.stream_copy(a.ptr, b.ptr, 17)
    a.close(); b.close()


@pytest.mark.parametrize("n_tables", [1, 2, 5, 9])
def test_awkward_table_values_and_fewer_tables(engine, n_tables):
    t = random_tables(40 + n_tables, 9)
    engine.set_tables(t[:n_tables])
    esa, gt, coarse, sgt = make_block(n_tables, 90, 333, 5, 15, nasty=True)
    mask = (1 << n_tables) - 1
    want = oc.process_block_mem(esa, gt, coarse, sgt, t, table_mask=mask)
    got = engine.process_block_mem(esa, gt, coarse, sgt, table_mask=mask)
    assert np.array_equal(got, want)
    with pytest.raises(gpu.Gcn10GpuError):          # table not loaded
        engine.process_block_mem(esa, gt, coarse, sgt, table_mask=1 << n_tables if n_tables < 9 else 1 << 9)


def test_golden_digests(eng9, tables):
    spec = json.load(open(os.path.join(GOLDEN, "oracle_blocks.json")))
    for case in spec["cases"]:
        esa, gt, coarse, sgt = make_block(case["seed"], case["H"], case["W"], case["hsy"],
                                          case["hsx"], nasty=case["nasty"])
        got = eng9.process_block_mem(esa, gt, coarse, sgt)
        for r in range(18):
            assert hashlib.sha256(got[r].tobytes()).hexdigest() == case["sha256"][r], (case["seed"], r)


def test_argument_errors(eng9):
    with pytest.raises(gpu.Gcn10GpuError) as e:
        eng9.cn_strip(0, 16, 1, 0, 3, 0x1FF, [None] * 18)
    assert e.value.code in (gpu.lib().gcn10_gpu_abi_version() * 0 - 1, -4)
    esa, gt, coarse, sgt = make_block(1, 4, 16, 2, 2)
    with pytest.raises(gpu.Gcn10GpuError):
        eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=0)
    with pytest.raises(gpu.Gcn10GpuError):
        eng9.process_block_mem(esa, gt, coarse, sgt, cond_mask=4)
    empty = eng9.process_block_mem(np.zeros((0, 16), np.uint8), gt, coarse, sgt)
    assert empty.shape == (18, 0, 16)


# ---- BASELINE.json full size: 36000 x 36000 --------------------------------

@pytest.fixture(scope="module")
def full_tile(engine, tables):
    """One 36000^2 block resident on the GPU with all 18 rasters computed once."""
    H = W = 36000
    rng = np.random.default_rng(1)
    # spatially coherent landcover (64-px patches) with noise, cheap to generate
    small = rng.choice(np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], np.uint8),
                       size=(H // 60, W // 60))
    esa = np.repeat(np.repeat(small, 60, axis=0), 60, axis=1)
    noise = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    esa = np.where(noise < 8, noise, esa).astype(np.uint8)
    del noise
    coarse = rng.choice(np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], np.uint8), size=(1440, 1440))
    gt = [0.0, 3.0 / W, 0.0, 3.0, 0.0, -3.0 / W]
    sgt = [0.0, 3.0 / 1440, 0.0, 3.0, 0.0, -3.0 / 1440]
    ci, cj = host.build_index_maps(gt, sgt, W, H, 1440, 1440)
    e = engine
    e.set_tables(tables)
    bufs = dict(esa=e.upload(esa), coarse=e.upload(coarse), ci=e.upload(ci), cj=e.upload(cj))
    outs = [e.alloc(H * W) for _ in range(18)]
    e.prepare_tile(bufs["coarse"].ptr, 1440, 1440, bufs["ci"].ptr, W)
    e.cn_strip(bufs["esa"].ptr, W, H, bufs["cj"].ptr, 3, 0x1FF, [o.ptr for o in outs])
    e.sync()
    yield dict(H=H, W=W, esa=esa, coarse=coarse, gt=gt, sgt=sgt, ci=ci, cj=cj, bufs=bufs, outs=outs)
    for b in list(bufs.values()) + outs:
        b.close()


def test_full_size_sampled_rows_equal_oracle(eng9, tables, full_tile):
    """Oracle on 48 sampled rows (first, last, around coarse-row changes) of the 36000^2 block."""
    ft = full_tile
    H, W = ft["H"], ft["W"]
    rng = np.random.default_rng(2)
    rows = sorted(set([0, 1, H - 1, H - 2, 12, 13, 37, 38] + rng.integers(0, H, 40).tolist()))
    fine_rows = ft["coarse"][ft["cj"][rows]][:, ft["ci"]]        # checked against oracle_resample below
    r0 = rows[5]
    sub_gt = list(ft["gt"]); sub_gt[3] = ft["gt"][3] + r0 * ft["gt"][5]
    assert np.array_equal(oc.resample(ft["coarse"], sub_gt, ft["sgt"], W, 1)[0], fine_rows[5])
    for c in range(2):
        adj = oc.modify_hysogs_data(fine_rows, drained=(c == 0))
        for k in range(9):
            want = oc.calculate_cn(ft["esa"][rows], adj, tables[k])
            r = c * 9 + k
            for i, y in enumerate(rows):
                got = eng9.download(ft["outs"][r].at(y * W), (W,))
                assert same(got, want[i], "raster %d row %d (soil rows %s)" % (r, y, ft["cj"][max(y - 1, 0):y + 2])), (r, y)


def test_full_size_histogram_identity(eng9, tables, full_tile):
    """Checksum of checksums: the histogram of a CN raster is fixed by the joint
    histogram of (landcover, resampled soil) and the table -- size independent."""
    ft = full_tile
    H, W = ft["H"], ft["W"]
    fine = ft["coarse"][ft["cj"][:, None], ft["ci"][None, :]]
    joint = np.bincount(ft["esa"].astype(np.int64).ravel() * 256 + fine.ravel(), minlength=65536)
    joint = joint.reshape(256, 256)
    del fine
    for r in (0, 7, 9 + 7, 17):
        c, k = divmod(r, 9)
        hmap = np.arange(256)
        dual = (hmap >= 11) & (hmap <= 14)
        s = np.where(dual, 4 if c == 0 else hmap - 10, hmap)
        t = tables[k]
        val = np.full((256, 256), 255, dtype=np.int64)
        ok = s < 5
        tv = t[:, s[ok]]
        val[:, ok] = np.where(tv < 255, tv & 255, 255)
        want = np.bincount(val.ravel(), weights=joint.ravel(), minlength=256).astype(np.int64)
        got_raster = eng9.download(ft["outs"][r].ptr, (H * W,))
        got = np.bincount(got_raster, minlength=256)
        assert np.array_equal(got, want), r


def test_full_size_strips_and_single_table_kernels_agree(eng9, full_tile):
    """Idempotence / decomposition: strips of 4000 rows on a side stream, and the
    single-table kernel, reproduce the whole-block launch byte for byte."""
    ft = full_tile
    H, W = ft["H"], ft["W"]
    e = eng9
    s = e.stream_create()
    tmp = e.alloc(H * W)
    for r in (7, 9 + 2):
        c, k = divmod(r, 9)
        e.memset(tmp.ptr, 0xA5, H * W, s)
        for y0 in range(0, H, 4000):
            ptrs = [None] * 18
            ptrs[r] = tmp.at(y0 * W)
            e.cn_strip(ft["bufs"]["esa"].at(y0 * W), W, 4000, ft["bufs"]["cj"].at(4 * y0),
                       1 << c, 1 << k, ptrs, s)
        e.sync(s)
        assert e.last_kernel_name().startswith("cn_strip_kernel<1,")
        a = e.download(tmp.ptr, (H * W,))
        b = e.download(ft["outs"][r].ptr, (H * W,))
        assert np.array_equal(a, b)
    tmp.close()
    e.stream_destroy(s)


def test_full_size_nontemporal_and_plain_stores_write_the_same_bytes(eng9, full_tile):
    """Pins the default store policy: a full-size raster written into a plain device
    allocation with nontemporal stores (the default of every product raster) and once more with plain stores,
    both read back right after the writer's stream has been synchronised, every byte compared -- for the
    single-raster kernel and for the 18-raster kernel (raster 0 of it).  A store that was not yet visible when
    the kernel had completed (the hazard seen with chunk-mapped ranges in round 2,
    profiles/r02/spread_allocator_hazard.txt) would leave the 0x5A / 0xA5 fill behind."""
    ft = full_tile
    H, W = ft["H"], ft["W"]
    e = eng9
    nt, plain = e.alloc(H * W), e.alloc(H * W)
    try:
        for kind in ("single", "all18"):
            e.memset(nt.ptr, 0x5A, H * W)
            e.memset(plain.ptr, 0xA5, H * W)
            e.sync()
            for buf, policy in ((nt, 1), (plain, 0)):
                e.set_option("nontemporal", policy)
                e.prepare_tile(ft["bufs"]["coarse"].ptr, 1440, 1440, ft["bufs"]["ci"].ptr, W)
                if kind == "single":
                    ptrs = [None] * 18
                    ptrs[9 + 4] = buf.ptr
                    e.cn_strip(ft["bufs"]["esa"].ptr, W, H, ft["bufs"]["cj"].ptr, 2, 1 << 4, ptrs)
                else:
                    # the other 17 rasters of the launch go where the fixture's rasters are (same bytes again)
                    ptrs = [o.ptr for o in ft["outs"]]
                    ptrs[0] = buf.ptr
                    e.cn_strip(ft["bufs"]["esa"].ptr, W, H, ft["bufs"]["cj"].ptr, 3, 0x1FF, ptrs)
                e.sync()
            ref = ft["outs"][9 + 4 if kind == "single" else 0]
            a = e.download(nt.ptr, (H * W,))
            b = e.download(plain.ptr, (H * W,))
            c = e.download(ref.ptr, (H * W,))
            assert np.array_equal(a, b), "%s: nontemporal and plain stores differ in %d bytes" % (
                kind, int(np.count_nonzero(a != b)))
            assert np.array_equal(a, c), kind
            del a, b, c
    finally:
        e.set_option("defaults", 0)
        nt.close()
        plain.close()

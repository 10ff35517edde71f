"""The oracle against itself and against what the reference's own files pin.

PARITY UNPINNED: the reference has no golden rasters / KATs for this path and
cannot be built in this image (oracle/README.md).  What CAN be pinned is pinned
here: (1) the shipped lookup CSVs are the per-(class, letter) answers, so every
CSV row is a known-answer vector for calculate_cn on unmodified soil groups;
(2) two independently written restatements (C loops, numpy algebra) and a third
table-algebra formulation must agree bit for bit on everything else.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import cn_oracle_c as oc
from oracle import cn_oracle_np as onp
from tests.conftest import GOLDEN, LOOKUPS
from tests.util import make_block, random_tables

LETTER = {"A": 1, "B": 2, "C": 3, "D": 4}


def _csv_rows(path):
    raw = open(path, "rb").read().decode("utf-8-sig").replace("\r", "")
    rows = []
    for line in raw.split("\n")[1:]:
        if not line:
            continue
        code, cn = line.split(",")
        lc, letter = code.split("_")
        rows.append((int(lc), letter, int(cn)))
    return rows


@pytest.mark.parametrize("hc", onp.HCS)
@pytest.mark.parametrize("arc", onp.ARCS)
def test_shipped_csv_rows_are_known_answers(hc, arc):
    path = os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc))
    rows = _csv_rows(path)
    assert len(rows) == 44                      # 11 classes x A-D
    tc, bad_c = oc.load_lookup_table(path)
    tn, bad_n = onp.load_lookup_table(path)
    assert bad_c == 0 and bad_n == 0
    assert np.array_equal(tc, tn)
    assert (tc != 255).sum() == 44 and (tc[:, 0] == 255).all()
    for lc, letter, cn in rows:
        sg = LETTER[letter]
        assert tc[lc, sg] == cn
        esa = np.array([[lc]], dtype=np.uint8)
        for fn in (oc.calculate_cn, onp.calculate_cn):
            # plain soil group: the row's value, both drainage conditions
            assert fn(esa, np.array([[sg]], dtype=np.uint8), tc)[0, 0] == cn
        # dual class 10+sg: undrained acts as the first letter (src/cn.c:101-108),
        # drained as D (src/cn.c:94-96)
        dual = np.array([[10 + sg]], dtype=np.uint8)
        und = oc.modify_hysogs_data(dual, drained=False)
        dr = oc.modify_hysogs_data(dual, drained=True)
        assert und[0, 0] == sg and dr[0, 0] == 4
        assert oc.calculate_cn(esa, und, tc)[0, 0] == cn
        assert oc.calculate_cn(esa, dr, tc)[0, 0] == tc[lc, 4]


def test_known_values_spot_check(tables):
    # lookups/default_lookup_g_ii.csv:2-5 -> 10_A..10_D = 15,35,51,59 ; k = g*3+ii = 7
    assert tables[7][10].tolist() == [255, 15, 35, 51, 59]
    # lookups/default_lookup_p_i.csv:2-5 -> 45,66,77,83 ; k = 0
    assert tables[0][10].tolist() == [255, 45, 66, 77, 83]


def test_modify_hysogs_all_codes():
    h = np.arange(256, dtype=np.uint8)
    for drained in (True, False):
        a = oc.modify_hysogs_data(h, drained)
        b = onp.modify_hysogs_data(h, drained)
        assert np.array_equal(a, b)
        exp = h.copy()
        exp[11:15] = 4 if drained else np.array([1, 2, 3, 4])
        assert np.array_equal(a, exp)


def test_calculate_cn_truncating_cast_and_nodata():
    t = np.full((256, 5), 255, dtype=np.int32)
    t[1, 1] = -1        # (uint8_t)-1 = 255 stored as a value (src/cn.c:126-127)
    t[2, 2] = 256       # >= 255: left at the memset 255 (src/cn.c:126, 289)
    t[3, 3] = 254
    t[4, 4] = -256      # wraps to 0
    t[5, 0] = 7         # column 0 is reachable through calculate_cn
    esa = np.array([[1, 2, 3, 4, 5, 1]], dtype=np.uint8)
    hsg = np.array([[1, 2, 3, 4, 0, 5]], dtype=np.uint8)
    exp = [255, 255, 254, 0, 7, 255]
    assert oc.calculate_cn(esa, hsg, t)[0].tolist() == exp
    assert onp.calculate_cn(esa, hsg, t)[0].tolist() == exp


MALFORMED = [
    # (name, file bytes, expected {(lc,sg): cn}, expected bad rows)
    ("lowercase_letter_is_D", b"grid_code,cn\n10_a,5\n", {(10, 4): 5}, 0),
    ("no_underscore", b"h\n10A,5\n20_B,6\n", {(20, 2): 6}, 1),
    ("missing_cn", b"h\n10_A\n10_B,7\n", {(10, 2): 7}, 1),
    ("blank_line_counts_as_bad", b"h\n\n10_C,8\n", {(10, 3): 8}, 1),
    ("crlf_bom", b"\xef\xbb\xbfgrid_code,cn\r\n10_A,15\r\n", {(10, 1): 15}, 0),
    ("lc_out_of_range", b"h\n256_A,1\n-1_B,2\n255_C,3\n", {(255, 3): 3}, 2),
    ("negative_and_big_cn", b"h\n1_A,-3\n2_B,300\n", {(1, 1): -3, (2, 2): 300}, 0),
    ("leading_commas", b"h\n,,3_C,9,extra\n", {(3, 3): 9}, 0),
    ("underscore_at_end", b"h\n7_,4\n", {(7, 4): 4}, 0),
    ("non_numeric_lc_is_zero", b"h\nxx_B,4\n", {(0, 2): 4}, 0),
    ("no_trailing_newline", b"h\n9_D,1", {(9, 4): 1}, 0),
    ("spaces", b"h\n 12_A, 34 \n", {(12, 1): 34}, 0),
    ("long_line_is_cut_at_127", b"h\n" + b"5_A,1" + b"x" * 130 + b"\n6_B,2\n",
     {(5, 1): 1, (6, 2): 2}, 1),
    ("last_wins", b"h\n10_A,1\n10_A,2\n", {(10, 1): 2}, 0),
]


@pytest.mark.parametrize("name,data,expect,nbad", MALFORMED, ids=[m[0] for m in MALFORMED])
def test_lookup_loader_edge_cases(tmp_path, name, data, expect, nbad):
    p = tmp_path / "t.csv"
    p.write_bytes(data)
    tc, bad_c = oc.load_lookup_table(str(p))
    tn, bad_n = onp.load_lookup_table(str(p))
    assert np.array_equal(tc, tn)
    assert bad_c == bad_n == nbad
    exp = np.full((256, 5), 255, dtype=np.int32)
    for (lc, sg), cn in expect.items():
        exp[lc, sg] = cn
    assert np.array_equal(tc, exp)


def test_lookup_loader_missing_and_empty(tmp_path):
    with pytest.raises(FileNotFoundError):
        oc.load_lookup_table(str(tmp_path / "nope.csv"))
    (tmp_path / "e.csv").write_bytes(b"")
    with pytest.raises(ValueError):
        oc.load_lookup_table(str(tmp_path / "e.csv"))
    with pytest.raises(ValueError):
        onp.load_lookup_table(str(tmp_path / "e.csv"))


def test_c_round_is_half_away_from_zero():
    v = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.49999999999999994, -0.49999999999999994,
                  1e15 + 0.5, 3.0, -3.0])
    got = onp._c_round(v)
    exp = np.array([1.0, 2.0, 3.0, -1.0, -2.0, 0.0, -0.0, 1e15 + 1, 3.0, -3.0])
    assert np.array_equal(got, exp)


def test_double_to_int_x86():
    L = oc.lib()
    assert L.oracle_double_to_int_x86(3.99) == 3
    assert L.oracle_double_to_int_x86(-3.99) == -3
    assert L.oracle_double_to_int_x86(2147483647.0) == 2147483647
    assert L.oracle_double_to_int_x86(2147483648.0) == -2**31
    assert L.oracle_double_to_int_x86(-2147483648.0) == -2**31
    assert L.oracle_double_to_int_x86(1e300) == -2**31
    assert L.oracle_double_to_int_x86(float("nan")) == -2**31
    assert L.oracle_double_to_int_x86(float("inf")) == -2**31


def test_index_maps_hand_derived():
    # fine 8 px over [0,1), coarse 2 cells over [0,1): dc = (x+0.5)/8 / 0.5 = (x+0.5)/4
    # -> 0.125, .375, .625, .875, 1.125, ... round -> 0,0,1,1,1,1,2->clamp 1,2->1
    gt = [0.0, 0.125, 0, 1.0, 0, -0.125]
    sgt = [0.0, 0.5, 0, 1.0, 0, -0.5]
    for fn in (oc.index_maps, onp.index_maps):
        ci, cj = fn(gt, sgt, 8, 8, 2, 2)
        assert ci.tolist() == [0, 0, 1, 1, 1, 1, 1, 1]
        assert cj.tolist() == [0, 0, 1, 1, 1, 1, 1, 1]
    # exact tie: dc = 0.5 rounds away from zero to 1 (src/cn.c:225), -0.5 to -1 -> clamp 0
    gt = [0.0, 1.0, 0, 0.0, 0, -1.0]
    sgt = [0.0, 1.0, 0, 0.0, 0, -1.0]          # dc = x + 0.5 -> x + 1
    ci, cj = oc.index_maps(gt, sgt, 4, 4, 10, 10)
    assert ci.tolist() == [1, 2, 3, 4] and cj.tolist() == [1, 2, 3, 4]
    sgt = [1.0, 1.0, 0, -1.0, 0, -1.0]         # dc = x - 0.5 -> -1(->0), 1, 2, 3
    ci, cj = oc.index_maps(gt, sgt, 4, 4, 10, 10)
    assert ci.tolist() == [0, 1, 2, 3] and cj.tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("seed", range(6))
def test_resample_c_vs_numpy(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    hsy, hsx = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    esa, gt, coarse, sgt = make_block(seed, H, W, hsy, hsx, nasty=True)
    a = oc.resample(coarse, gt, sgt, W, H)
    b = onp.resample(coarse, gt, sgt, W, H)
    assert np.array_equal(a, b)
    ci, cj = oc.index_maps(gt, sgt, W, H, hsx, hsy)
    ci2, cj2 = onp.index_maps(gt, sgt, W, H, hsx, hsy)
    assert np.array_equal(ci, ci2) and np.array_equal(cj, cj2)
    assert np.array_equal(a, coarse[cj[:, None], ci[None, :]])   # separability


def test_resample_degenerate_geotransforms():
    coarse = np.arange(12, dtype=np.uint8).reshape(3, 4)
    gt = [0.0, 1.0, 0, 0.0, 0, -1.0]
    # soil grid far away: everything clamps to an edge
    for sgt in ([1e6, 1.0, 0, 0.0, 0, -1.0], [-1e6, 1.0, 0, 1e6, 0, -1.0],
                [0.0, 1e-12, 0, 0.0, 0, -1e-12],      # dc overflows int -> INT_MIN -> 0
                [0.0, 0.0, 0, 0.0, 0, 0.0]):          # division by zero -> inf/nan -> 0
        a = oc.resample(coarse, gt, sgt, 5, 5)
        b = onp.resample(coarse, gt, sgt, 5, 5)
        assert np.array_equal(a, b)


VRT_GT = [-180.0, 8.3333333333330430e-05, 0.0, 84.0, 0.0, -8.3333333333330430e-05]  # landcover/esa_worldcover_2021.vrt:3


def test_window_real_vrt_geotransform():
    # a 3x3 degree block on integer degrees gives 36001 px
    for fn in (oc.window, onp.window):
        xo, yo, xc, yc, gt = fn(VRT_GT, 4320000, 1728000, [-111.0, 36.0, -108.0, 39.0])
        assert (xc, yc) == (36001, 36001)
        assert xo == 828000 and yo == 540000
        assert gt[1] == VRT_GT[1] and gt[5] == VRT_GT[5]
    # the raster's east / south edge clamps to 36000
    xo, yo, xc, yc, _ = oc.window(VRT_GT, 4320000, 1728000, [177.0, -60.0, 180.0, -57.0])
    assert xo + xc == 4320000 and xc == 36000
    # outside
    assert oc.window(VRT_GT, 4320000, 1728000, [200.0, 0.0, 203.0, 3.0]) is None
    assert onp.window(VRT_GT, 4320000, 1728000, [200.0, 0.0, 203.0, 3.0]) is None
    # negative offsets shrink the count (src/raster.c:134-141)
    a = oc.window(VRT_GT, 4320000, 1728000, [-181.0, 83.0, -179.0, 85.0])
    b = onp.window(VRT_GT, 4320000, 1728000, [-181.0, 83.0, -179.0, 85.0])
    assert a == b and a[0] == 0 and a[1] == 0


@pytest.mark.parametrize("seed", range(8))
def test_window_c_vs_numpy_random(seed):
    rng = np.random.default_rng(100 + seed)
    t = [float(rng.uniform(-180, 0)), float(rng.uniform(1e-4, 1e-2)), 0.0,
         float(rng.uniform(0, 84)), 0.0, -float(rng.uniform(1e-4, 1e-2))]
    rx, ry = int(rng.integers(10, 5000)), int(rng.integers(10, 5000))
    for _ in range(20):
        x0 = t[0] + float(rng.uniform(-5, rx + 5)) * t[1]
        y1 = t[3] + float(rng.uniform(-5, ry + 5)) * t[5]
        bbox = [x0, y1 - float(rng.uniform(0, 3)), x0 + float(rng.uniform(0, 3)), y1]
        assert oc.window(t, rx, ry, bbox) == onp.window(t, rx, ry, bbox)


@pytest.mark.parametrize("shape", [(1, 1), (1, 17), (17, 1), (5, 3), (64, 64), (257, 131), (300, 300)])
def test_process_block_three_formulations_agree(tables, shape):
    H, W = shape
    esa, gt, coarse, sgt = make_block(7 + H * W, H, W, max(1, H // 25 + 2), max(1, W // 25 + 2), nasty=True)
    a = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    b = onp.process_block_mem(esa, gt, coarse, sgt, tables)
    c = onp.fused_semantics(esa, onp.resample(coarse, gt, sgt, W, H), tables)
    assert np.array_equal(a, b)
    assert np.array_equal(a, c)


def test_process_block_random_tables_and_subset():
    t = random_tables(3)
    esa, gt, coarse, sgt = make_block(11, 40, 33, 4, 5, nasty=True)
    a = oc.process_block_mem(esa, gt, coarse, sgt, t)
    b = onp.process_block_mem(esa, gt, coarse, sgt, t)
    assert np.array_equal(a, b)
    sub = oc.process_block_mem(esa, gt, coarse, sgt, t, cond_mask=2, table_mask=0b100000001)
    assert np.array_equal(sub[9], a[9]) and np.array_equal(sub[17], a[17])
    assert not sub[:9].any() and not sub[10:17].any()
    assert oc.process_block_mem(esa, gt, coarse, sgt, t, want_output=False) is None


def test_exhaustive_pairs_tile(tables):
    """esa[y][x] = y, resampled HSG = x: all 65536 (class, soil code) pairs x 18 rasters."""
    esa = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 256, axis=1)
    coarse = np.arange(256, dtype=np.uint8)[None, :]
    gt = [0.0, 1.0, 0.0, 0.0, 0.0, -1.0]
    sgt = [0.5, 1.0, 0.0, 0.0, 0.0, -1.0]       # dc = x exactly
    fine = oc.resample(coarse, gt, sgt, 256, 256)
    assert np.array_equal(fine, np.repeat(coarse, 256, axis=0))
    out = oc.process_block_mem(esa, gt, coarse, sgt, tables)
    assert np.array_equal(out, onp.process_block_mem(esa, gt, coarse, sgt, tables))
    # direct statement of the rule for this tile
    for c in range(2):
        for k in range(9):
            for h in (0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15, 255):
                s = h
                if 11 <= h <= 14:
                    s = 4 if c == 0 else h - 10
                col = out[c * 9 + k][:, h]
                if s < 5:
                    v = tables[k][:, s]
                    assert np.array_equal(col, np.where(v < 255, v & 255, 255).astype(np.uint8))
                else:
                    assert (col == 255).all()


def test_golden_fixture_regression(tables):
    """tests/golden/oracle_blocks.json: SHA-256 of oracle outputs on seeded blocks.

    These digests were produced by THIS oracle (tests/golden/make_golden.py), not
    by the reference (which cannot run here): they freeze the oracle against
    accidental edits, they do not pin it to the reference."""
    spec = json.load(open(os.path.join(GOLDEN, "oracle_blocks.json")))
    for case in spec["cases"]:
        esa, gt, coarse, sgt = make_block(case["seed"], case["H"], case["W"], case["hsy"],
                                          case["hsx"], nasty=case["nasty"])
        out = oc.process_block_mem(esa, gt, coarse, sgt, tables)
        for r in range(18):
            assert hashlib.sha256(out[r].tobytes()).hexdigest() == case["sha256"][r], (case, r)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (37, 131, 3, 7), (64, 1041, 5, 44), (3, 5000, 2, 201)])
def test_fused_best_cpu_pass_equals_reference_shaped_pass(shape):
    """oracle/cn_fused_cpu.c (bench.py's cpu_baseline.best_cpu line, built -march=native on the host it
    runs on) against oracle_process_block_subset, byte for byte, incl. awkward table values and subsets."""
    H, W, hy, hx = shape
    t = random_tables(3, 9)
    esa, gt, coarse, sgt = make_block(H * 31 + W, H, W, hy, hx, nasty=True)
    for cm, tm in [(3, 0x1FF), (1, 0x80), (2, 0x0A5)]:
        a = oc.process_block_mem(esa, gt, coarse, sgt, t, cond_mask=cm, table_mask=tm)
        b = oc.fused_block(esa, gt, coarse, sgt, t, cond_mask=cm, table_mask=tm)
        assert np.array_equal(a, b), (shape, cm, tm)

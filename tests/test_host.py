"""libgcn10_host.so (product, C99) against the oracle: lookup loader and fp64 geo math."""
import os

import numpy as np
import pytest

from gcn10_amd import host
from oracle import cn_oracle_c as oc
from tests.conftest import LOOKUPS
from tests.test_oracle import MALFORMED, VRT_GT
from tests.util import make_block


def test_shipped_tables_equal_oracle(tables):
    got = host.load_all_lookup_tables(LOOKUPS)
    assert got.dtype == np.int32 and got.shape == (9, 256, 5)
    assert np.array_equal(got, tables)


@pytest.mark.parametrize("name,data,expect,nbad", MALFORMED, ids=[m[0] for m in MALFORMED])
def test_lookup_edge_cases_equal_oracle(tmp_path, name, data, expect, nbad):
    p = tmp_path / "t.csv"
    p.write_bytes(data)
    want, bad = oc.load_lookup_table(str(p))
    got, msgs = host.load_lookup_file(str(p))
    assert np.array_equal(got, want)
    assert len(msgs) == bad == nbad


def test_lookup_error_messages_match_reference_text(tmp_path):
    p = tmp_path / "t.csv"
    p.write_bytes(b"h\n10A,5\n10_A\n300_B,2\n")
    _, msgs = host.load_lookup_file(str(p))
    assert msgs[0] == "invalid grid_code in %s: 10A" % p          # src/cn.c:59-60
    assert msgs[1] == "invalid row in %s: missing cn" % p          # src/cn.c:69-70
    assert msgs[2] == "invalid values in %s: lc=300, sg=2" % p     # src/cn.c:79-80


def test_lookup_missing_and_empty(tmp_path):
    with pytest.raises(host.LookupError_, match="cannot open lookup table"):
        host.load_lookup_file(str(tmp_path / "missing.csv"))
    (tmp_path / "e.csv").write_bytes(b"")
    with pytest.raises(host.LookupError_, match="empty lookup table"):
        host.load_lookup_file(str(tmp_path / "e.csv"))
    with pytest.raises(host.LookupError_):
        host.load_all_lookup_tables(str(tmp_path))


@pytest.mark.parametrize("seed", range(10))
def test_index_maps_equal_oracle(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    hsy, hsx = int(rng.integers(1, 130)), int(rng.integers(1, 130))
    _, gt, _, sgt = make_block(seed, H, W, hsy, hsx)
    ci, cj = host.build_index_maps(gt, sgt, W, H, hsx, hsy)
    ci0, cj0 = oc.index_maps(gt, sgt, W, H, hsx, hsy)
    assert np.array_equal(ci, ci0) and np.array_equal(cj, cj0)
    assert ci.min() >= 0 and ci.max() < hsx and cj.min() >= 0 and cj.max() < hsy


def test_index_maps_full_size_real_pixel():
    # the real block shape: 36001 px of 8.333e-05 deg over a 250 m (1/480 deg) soil grid
    px = VRT_GT[1]
    gt = [-111.0000000000024, px, 0.0, 39.00000000000157, 0.0, -px]
    sgt = [-111.00208333333, 1.0 / 480.0, 0.0, 39.00208333333, 0.0, -1.0 / 480.0]
    ci, cj = host.build_index_maps(gt, sgt, 36001, 36001, 1442, 1442)
    ci0, cj0 = oc.index_maps(gt, sgt, 36001, 36001, 1442, 1442)
    assert np.array_equal(ci, ci0) and np.array_equal(cj, cj0)
    assert np.all(np.diff(ci) >= 0) and np.all(np.diff(cj) >= 0)


def test_index_maps_degenerate():
    gt = [0.0, 1.0, 0, 0.0, 0, -1.0]
    for sgt in ([1e6, 1.0, 0, 0.0, 0, -1.0], [0.0, 1e-12, 0, 0.0, 0, -1e-12],
                [0.0, 0.0, 0, 0.0, 0, 0.0], [0.0, -1.0, 0, 0.0, 0, 1.0]):
        a = host.build_index_maps(gt, sgt, 7, 5, 4, 3)
        b = oc.index_maps(gt, sgt, 7, 5, 4, 3)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_window_equal_oracle():
    cases = [[-111.0, 36.0, -108.0, 39.0], [177.0, -60.0, 180.0, -57.0],
             [200.0, 0.0, 203.0, 3.0], [-181.0, 83.0, -179.0, 85.0], [0.0, 0.0, 0.0, 0.0]]
    for bbox in cases:
        assert host.raster_window(VRT_GT, 4320000, 1728000, bbox) == \
            oc.window(VRT_GT, 4320000, 1728000, bbox)
    rng = np.random.default_rng(5)
    t = [-20.0, 0.0021, 0.0, 60.0, 0.0, -0.0019]
    for _ in range(200):
        x0 = t[0] + float(rng.uniform(-5, 1005)) * t[1]
        y1 = t[3] + float(rng.uniform(-5, 805)) * t[5]
        bbox = [x0, y1 - float(rng.uniform(0, 1)), x0 + float(rng.uniform(0, 1)), y1]
        assert host.raster_window(t, 1000, 800, bbox) == oc.window(t, 1000, 800, bbox)

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
LOOKUPS = os.path.join(GOLDEN, "lookups")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build whatever is missing (in-tree, gfx950 cross-compile works without a GPU)."""
    import gcn10_amd
    from gcn10_amd import gpu, host
    need = []
    if not os.path.exists(gpu.LIB_PATH):
        need.append("gpu")
    if not os.path.exists(host.LIB_PATH):
        need.append("host")
    if not os.path.exists(os.path.join(ROOT, "oracle", "libcn_oracle.so")):
        need.append("oracle")
    if not os.path.exists(os.path.join(ROOT, "bin", "gcn10")) and \
            os.path.exists(os.path.join(ROOT, "gcn10_amd", "csrc", "host", "main.c")):
        need.append("cli")
    if need:
        gcn10_amd.build_all(tuple(need))


@pytest.fixture(scope="session")
def lookups_dir():
    return LOOKUPS


@pytest.fixture(scope="session")
def tables(_native_built):
    """The nine shipped lookup tables through the ORACLE's loader, int32[9,256,5]."""
    from oracle import cn_oracle_c as oc
    from oracle import cn_oracle_np as onp
    out = []
    for hc in onp.HCS:
        for arc in onp.ARCS:
            t, bad = oc.load_lookup_table(os.path.join(LOOKUPS, "default_lookup_%s_%s.csv" % (hc, arc)))
            assert bad == 0
            out.append(t)
    return np.stack(out)


@pytest.fixture(scope="session")
def engine(_native_built):
    from gcn10_amd import gpu
    eng = gpu.Engine(0)
    yield eng
    eng.close()

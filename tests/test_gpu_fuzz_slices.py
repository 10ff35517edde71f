"""Short seeded slices of the two randomised codec checks, so that every round's GPU run covers them
(the long runs are `python tests/fuzz_fused_encoder.py --cases N` and `python tools/fuzz_inflate.py
--streams N`; their latest results are under profiles/)."""
import importlib.util
import os

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [20261004, 7])
def test_fused_encoder_random_blocks(seed):
    """Random shapes / textures / soil windows / tables / raster subsets: every stream of the fused tile
    encoder inflates (stock zlib) to the oracle's tile."""
    fz = _load(os.path.join(ROOT, "tests", "fuzz_fused_encoder.py"), "fuzz_fused_encoder")
    cases, streams, diff = fz.run(seed=seed, cases=60, verbose=False)
    assert diff is None, diff
    assert cases == 60 and streams > 500


@pytest.mark.parametrize("seed", [20261004, 11])
def test_inflate_random_streams(seed):
    """Random zlib streams of every level / strategy / window / flush pattern, 15 % of them corrupted:
    good ones decode byte for byte, corrupted ones are refused or decode as zlib decodes them."""
    fz = _load(os.path.join(ROOT, "tools", "fuzz_inflate.py"), "fuzz_inflate")
    n, ok, refused, bad = fz.run(seed=seed, streams_budget=512, batch=256, verbose=False)
    assert not bad, bad[:5]
    assert n == 512 and ok + refused == 512 and refused >= 1        # a few of the ~75 corrupted streams are refused

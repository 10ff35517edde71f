"""examples/reference_binding.c -- the reference-side binding of INTEGRATION.md section 2 -- compiles
against the two public headers with plain gcc, links the two libraries, and computes what the
oracle computes."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import LOOKUPS, ROOT
from tests.util import make_block

SRC = os.path.join(ROOT, "examples", "reference_binding.c")


def _compile(tmp_path, main=True):
    exe = str(tmp_path / "reference_binding")
    cmd = ["gcc", "-std=c99", "-D_GNU_SOURCE", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include")]
    if main:
        cmd += ["-DGCN10_BINDING_MAIN", SRC, "-o", exe, "-L" + os.path.join(ROOT, "gcn10_amd"), "-lgcn10_gpu",
                "-lgcn10_host", "-Wl,-rpath," + os.path.join(ROOT, "gcn10_amd")]
    else:
        cmd += ["-c", SRC, "-o", str(tmp_path / "reference_binding.o")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_binding_example_compiles_against_the_public_headers(tmp_path, _native_built):
    _compile(tmp_path, main=False)
    _compile(tmp_path, main=True)


@pytest.mark.gpu
def test_binding_example_equals_oracle(tmp_path, tables):
    from oracle import cn_oracle_c as oc
    exe = _compile(tmp_path)
    H, W = 333, 1021
    esa, gt, coarse, sgt = make_block(17, H, W, H // 25 + 2, W // 25 + 2, nasty=True)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.asarray(gt, np.float64).tobytes())
        f.write(np.asarray(sgt, np.float64).tobytes())
        f.write(esa.tobytes())
        f.write(coarse.tobytes())
    out = subprocess.run([exe, LOOKUPS, str(W), str(H), str(coarse.shape[1]), str(coarse.shape[0]),
                          str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(tmp_path / "out.bin", np.uint8).reshape(18, H, W)
    assert np.array_equal(got, oc.process_block_mem(esa, gt, coarse, sgt, tables))

"""The C-ABI libraries load and export every symbol include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from gcn10_amd import gpu, host
from tests.conftest import ROOT

DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(gcn10_\w+)\s*\(", re.M)


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(DECL.findall(text))
    # typedef'd function-pointer types are not symbols
    return {n for n in names if not n.endswith("_fn")}


def test_gpu_library_exports_every_declared_symbol():
    declared = _declared("gcn10_gpu.h")
    assert declared == set(gpu.ABI_SYMBOLS)
    L = gpu.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.gcn10_gpu_abi_version() == 3


def test_host_library_exports_every_declared_symbol():
    declared = _declared("gcn10_host.h")
    assert declared, "no declarations parsed"
    L = host.lib()
    for name in declared:
        getattr(L, name)        # raises AttributeError if the symbol is missing


def test_gpu_header_is_plain_c(tmp_path):
    """The boundary must compile as C99 with no HIP / C++ in sight."""
    src = tmp_path / "t.c"
    src.write_text('#include "gcn10_gpu.h"\n#include "gcn10_host.h"\nint main(void){return GCN10_N_RASTERS-18;}\n')
    rc = os.system("gcc -std=c99 -Wall -Werror -pedantic -I%s/include -c %s -o %s"
                   % (ROOT, src, tmp_path / "t.o"))
    assert rc == 0


def test_algorithmic_bytes_formula():
    # 19.0016 B/px for all 18 rasters of a 36000^2 block
    b = gpu.strip_algorithmic_bytes(36000, 36000, 1440, 1440, 3, 0x1FF)
    assert b == 36000 * 36000 * 19 + 1440 * 1440 + 4 * 72000
    assert abs(b / 36000 ** 2 - 19.0016) < 3e-4
    assert gpu.strip_algorithmic_bytes(36000, 36000, 1440, 1440, 1, 1 << 7) == \
        36000 * 36000 * 2 + 1440 * 1440 + 4 * 72000
    assert gpu.strip_algorithmic_bytes(0, 10, 1, 1, 3, 1) == 0


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the engine must refuse to start, loudly."""
    if gpu.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gpu.Gcn10GpuError) as e:
        gpu.Engine(0)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """Nothing under gcn10_amd/ or include/ (nor bench's GPU leg) may reference oracle/."""
    bad = []
    for base in ("gcn10_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"\boracle\b", text) and "cn_oracle" in text:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    out = os.popen("ldd %s %s 2>/dev/null" % (gpu.LIB_PATH, host.LIB_PATH)).read()
    assert "cn_oracle" not in out

"""gcn10_amd -- MI355X-native curve-number (CN) raster generator.

Python here is plumbing for tests and benchmarks: thin ctypes faces of the two
native libraries that make up the product,

* ``libgcn10_gpu.so``  -- hand-written gfx950 HIP kernels behind the C ABI of
  ``include/gcn10_gpu.h`` (:mod:`gcn10_amd.gpu`);
* ``libgcn10_host.so`` -- the C99 host side, ``include/gcn10_host.h``
  (:mod:`gcn10_amd.host`); the ``bin/gcn10`` program links the same code.

There is no CPU implementation of the per-pixel path in this package: without
the HIP library and a gfx950 device every compute call raises.
"""
from ._build import build_all, repo_root  # noqa: F401

__version__ = "0.1.0"

"""ctypes face of ``libgcn10_host.so`` (``include/gcn10_host.h``)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgcn10_host.so")
_lib = None

HCS = ("p", "f", "g")               # src/cn.c:146
ARCS = ("i", "ii", "iii")           # src/cn.c:147
CONDS = ("drained", "undrained")    # src/cn.c:145

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
ROW_ERROR_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `make host` (or __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gcn10_load_lookup_file.argtypes = [C.c_char_p, _i32p, ROW_ERROR_FN, C.c_void_p]
        L.gcn10_load_lookup_file.restype = C.c_int
        L.gcn10_load_lookup_table.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, _i32p,
                                              ROW_ERROR_FN, C.c_void_p]
        L.gcn10_load_lookup_table.restype = C.c_int
        L.gcn10_load_all_lookup_tables.argtypes = [C.c_char_p, _i32p, C.POINTER(C.c_int),
                                                   ROW_ERROR_FN, C.c_void_p]
        L.gcn10_load_all_lookup_tables.restype = C.c_int
        L.gcn10_build_index_maps.argtypes = [_f64p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             _i32p, _i32p]
        L.gcn10_build_index_maps.restype = None
        L.gcn10_raster_window.argtypes = [_f64p, C.c_int, C.c_int, _f64p, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), _f64p]
        L.gcn10_raster_window.restype = C.c_int
        _lib = L
    return _lib


def _f(v, n):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(n))


class LookupError_(RuntimeError):
    pass


def load_lookup_file(path: str):
    """One CSV -> (int32[256,5] table, [messages of rejected rows])."""
    table = np.empty((256, 5), dtype=np.int32)
    msgs = []
    cb = ROW_ERROR_FN(lambda _u, m: msgs.append(m.decode(errors="replace")))
    rc = lib().gcn10_load_lookup_file(os.fsencode(path), table, cb, None)
    if rc == -1:
        raise LookupError_("cannot open lookup table %s" % path)       # src/cn.c:30
    if rc == -2:
        raise LookupError_("empty lookup table %s" % path)             # src/cn.c:44
    return table, msgs


def load_all_lookup_tables(lookup_dir: str) -> np.ndarray:
    """The nine tables of a run, int32[9,256,5], k = hc*3 + arc."""
    tables = np.empty((9, 256, 5), dtype=np.int32)
    failed = C.c_int(-1)
    cb = ROW_ERROR_FN(lambda _u, m: None)
    rc = lib().gcn10_load_all_lookup_tables(os.fsencode(lookup_dir), tables.reshape(-1),
                                            C.byref(failed), cb, None)
    if rc != 0:
        k = failed.value
        raise LookupError_("lookup table %s_%s in %s: error %d" %
                           (HCS[k // 3], ARCS[k % 3], lookup_dir, rc))
    return tables


def build_index_maps(gt, soil_gt, W: int, H: int, hsx: int, hsy: int):
    ci = np.empty(W, dtype=np.int32)
    cj = np.empty(H, dtype=np.int32)
    lib().gcn10_build_index_maps(_f(gt, 6), _f(soil_gt, 6), W, H, hsx, hsy, ci, cj)
    return ci, cj


def raster_window(t, rx: int, ry: int, bbox):
    """src/raster.c:126-162 -> (xoff, yoff, xcount, ycount, gt) or None."""
    xo, yo, xc, yc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    gt = np.empty(6, dtype=np.float64)
    rc = lib().gcn10_raster_window(_f(t, 6), rx, ry, _f(bbox, 4), C.byref(xo), C.byref(yo),
                                   C.byref(xc), C.byref(yc), gt)
    if rc != 0:
        return None
    return xo.value, yo.value, xc.value, yc.value, gt.tolist()

"""ctypes face of ``libcn_oracle.so`` (the C restatement, ``cn_oracle.c``).

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see ``cn_oracle.h``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcn_oracle.so")
_lib = None

_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libcn_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_load_lookup_table.argtypes = [C.c_char_p, _i32p, C.POINTER(C.c_int)]
        L.oracle_load_lookup_table.restype = C.c_int
        L.oracle_modify_hysogs_data.argtypes = [_u8p, C.c_int, C.c_int]
        L.oracle_modify_hysogs_data.restype = None
        L.oracle_calculate_cn.argtypes = [_u8p, _u8p, C.c_int, _i32p, _u8p]
        L.oracle_calculate_cn.restype = None
        L.oracle_resample.argtypes = [_u8p, C.c_int, C.c_int, _f64p, _f64p,
                                      C.c_int, C.c_int, _u8p]
        L.oracle_resample.restype = None
        L.oracle_index_maps.argtypes = [_f64p, _f64p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _i32p, _i32p]
        L.oracle_index_maps.restype = None
        L.oracle_window.argtypes = [_f64p, C.c_int, C.c_int, _f64p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), _f64p]
        L.oracle_window.restype = C.c_int
        L.oracle_process_block_subset.argtypes = [
            _u8p, C.c_int, C.c_int, _f64p, _u8p, C.c_int, C.c_int, _f64p, _i32p,
            C.c_uint, C.c_uint, C.POINTER(C.c_void_p)]
        L.oracle_process_block_subset.restype = C.c_int
        L.oracle_double_to_int_x86.argtypes = [C.c_double]
        L.oracle_double_to_int_x86.restype = C.c_int
        _lib = L
    return _lib


def _f6(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(6))


def load_lookup_table(path: str):
    table = np.empty((256, 5), dtype=np.int32)
    bad = C.c_int(0)
    rc = lib().oracle_load_lookup_table(os.fsencode(path), table, C.byref(bad))
    if rc == -1:
        raise FileNotFoundError(path)
    if rc == -2:
        raise ValueError("empty lookup table %s" % path)
    return table, bad.value


def modify_hysogs_data(h: np.ndarray, drained: bool) -> np.ndarray:
    out = np.ascontiguousarray(h, dtype=np.uint8).copy()
    lib().oracle_modify_hysogs_data(out.reshape(-1), out.size, int(bool(drained)))
    return out


def calculate_cn(esa: np.ndarray, hsg: np.ndarray, table: np.ndarray) -> np.ndarray:
    esa = np.ascontiguousarray(esa, dtype=np.uint8)
    hsg = np.ascontiguousarray(hsg, dtype=np.uint8)
    out = np.full(esa.shape, 255, dtype=np.uint8)       # src/cn.c:289
    lib().oracle_calculate_cn(esa.reshape(-1), hsg.reshape(-1), esa.size,
                              np.ascontiguousarray(table, dtype=np.int32), out.reshape(-1))
    return out


def resample(coarse: np.ndarray, gt, soil_gt, esax: int, esay: int) -> np.ndarray:
    coarse = np.ascontiguousarray(coarse, dtype=np.uint8)
    hsy, hsx = coarse.shape
    out = np.empty((esay, esax), dtype=np.uint8)
    lib().oracle_resample(coarse.reshape(-1), hsx, hsy, _f6(gt), _f6(soil_gt),
                          esax, esay, out.reshape(-1))
    return out


def index_maps(gt, soil_gt, esax: int, esay: int, hsx: int, hsy: int):
    ci = np.empty(esax, dtype=np.int32)
    cj = np.empty(esay, dtype=np.int32)
    lib().oracle_index_maps(_f6(gt), _f6(soil_gt), esax, esay, hsx, hsy, ci, cj)
    return ci, cj


def window(t, rx: int, ry: int, bbox):
    xo, yo, xc, yc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    gt = np.empty(6, dtype=np.float64)
    bb = np.ascontiguousarray(np.asarray(bbox, dtype=np.float64).reshape(4))
    rc = lib().oracle_window(_f6(t), rx, ry, bb, C.byref(xo), C.byref(yo),
                             C.byref(xc), C.byref(yc), gt)
    if rc != 0:
        return None
    return xo.value, yo.value, xc.value, yc.value, gt.tolist()


def process_block_mem(esa, gt, coarse, soil_gt, tables, cond_mask=3, table_mask=0x1FF,
                      want_output=True):
    """Reference-shaped block pass.  Returns uint8[18,H,W] (unselected rasters
    are left at 0) or, with want_output=False, None (timing runs)."""
    esa = np.ascontiguousarray(esa, dtype=np.uint8)
    coarse = np.ascontiguousarray(coarse, dtype=np.uint8)
    tables = np.ascontiguousarray(tables, dtype=np.int32).reshape(9, 256, 5)
    esay, esax = esa.shape
    hsy, hsx = coarse.shape
    ptrs = (C.c_void_p * 18)()
    out = None
    if want_output:
        out = np.zeros((18, esay, esax), dtype=np.uint8)
        for i in range(18):
            if (cond_mask >> (i // 9)) & 1 and (table_mask >> (i % 9)) & 1:
                ptrs[i] = out[i].ctypes.data
    rc = lib().oracle_process_block_subset(
        esa.reshape(-1), esax, esay, _f6(gt), coarse.reshape(-1), hsx, hsy,
        _f6(soil_gt), tables.reshape(-1), cond_mask, table_mask, ptrs)
    if rc != 0:
        raise MemoryError("oracle_process_block_subset")
    return out


# ---- the fused "best CPU" pass (cn_fused_cpu.c, built -march=native for the host it runs on) ----
_NATIVE_PATH = os.path.join(_HERE, "libcn_fused_native.so")
_native = None


def build_native(force: bool = True) -> str:
    """(Re)build libcn_fused_native.so HERE: -march=native code built on another machine may not run."""
    if force or not os.path.exists(_NATIVE_PATH):
        subprocess.run(["make", "-C", _HERE, "-B", "libcn_fused_native.so"], check=True, capture_output=True)
    return _NATIVE_PATH


def fused_block(esa, gt, coarse, soil_gt, tables, cond_mask=3, table_mask=0x1FF, want_output=True, _out=None):
    """One fused pass over a block: the same rasters as process_block_mem (index maps from
    oracle_index_maps).  `_out`: a list of 18 uint8[H,W] arrays (None where a raster is not selected)
    to write into instead of a fresh uint8[18,H,W]; with want_output=False nothing is returned."""
    global _native
    if _native is None:
        build_native(force=not os.path.exists(_NATIVE_PATH))
        N = C.CDLL(_NATIVE_PATH)
        N.oracle_fused_block.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, _i32p, _i32p, _i32p,
                                         C.c_uint, C.c_uint, C.POINTER(C.c_void_p)]
        N.oracle_fused_block.restype = C.c_int
        _native = N
    esa = np.ascontiguousarray(esa, dtype=np.uint8)
    coarse = np.ascontiguousarray(coarse, dtype=np.uint8)
    tables = np.ascontiguousarray(tables, dtype=np.int32).reshape(9, 256, 5)
    H, W = esa.shape
    hsy, hsx = coarse.shape
    ci, cj = index_maps(gt, soil_gt, W, H, hsx, hsy)
    out = _out if _out is not None else np.zeros((18, H, W), dtype=np.uint8)
    ptrs = (C.c_void_p * 18)()
    for i in range(18):
        if (cond_mask >> (i // 9)) & 1 and (table_mask >> (i % 9)) & 1:
            assert out[i] is not None and out[i].shape == (H, W) and out[i].flags.c_contiguous
            ptrs[i] = out[i].ctypes.data
    rc = _native.oracle_fused_block(esa.reshape(-1), W, H, coarse.reshape(-1), hsx, ci, cj, tables.reshape(-1),
                                    cond_mask, table_mask, ptrs)
    if rc != 0:
        raise MemoryError("oracle_fused_block")
    return out if want_output else None

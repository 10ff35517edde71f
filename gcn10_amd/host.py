"""ctypes face of ``libgcn10_host.so`` (``include/gcn10_host.h``)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCN10_HOST_LIB") or os.path.join(_HERE, "libgcn10_host.so")
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
        vp, ip, cp = C.c_void_p, C.POINTER(C.c_int), C.c_char_p
        L.gcn10_config_parse.argtypes = [cp, C.POINTER(Config), cp, C.c_size_t]
        L.gcn10_config_parse.restype = C.c_int
        L.gcn10_config_free.argtypes = [C.POINTER(Config)]
        L.gcn10_config_free.restype = None
        L.gcn10_log_open.argtypes = [cp, C.c_int]
        L.gcn10_log_open.restype = vp
        L.gcn10_log_message.argtypes = [vp, cp, cp, C.c_bool]
        L.gcn10_log_message.restype = None
        L.gcn10_log_close.argtypes = [vp]
        L.gcn10_log_close.restype = None
        L.gcn10_read_block_list.argtypes = [cp, ip]
        L.gcn10_read_block_list.restype = C.POINTER(C.c_int)
        L.gcn10_blocks_open.argtypes = [cp, C.POINTER(Blocks), cp, C.c_size_t]
        L.gcn10_blocks_open.restype = C.c_int
        L.gcn10_blocks_free.argtypes = [C.POINTER(Blocks)]
        L.gcn10_blocks_free.restype = None
        L.gcn10_blocks_find.argtypes = [C.POINTER(Blocks), C.c_int]
        L.gcn10_blocks_find.restype = C.c_int
        L.gcn10_raster_open.argtypes = [cp, cp, cp, C.c_size_t]
        L.gcn10_raster_open.restype = vp
        L.gcn10_raster_close.argtypes = [vp]
        L.gcn10_raster_close.restype = None
        L.gcn10_raster_info.argtypes = [vp, ip, ip, _f64p]
        L.gcn10_raster_info.restype = None
        L.gcn10_raster_read.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, cp, C.c_size_t]
        L.gcn10_raster_read.restype = C.c_int
        L.gcn10_raster_georef.argtypes = [vp]
        L.gcn10_raster_georef.restype = vp
        L.gcn10_save_raster.argtypes = [vp, C.c_int, C.c_int, _f64p, vp, cp, C.c_int, cp, C.c_size_t]
        L.gcn10_save_raster.restype = C.c_int
        L.free_ = C.CDLL(None).free
        L.free_.argtypes = [vp]
        L.free_.restype = None
        _lib = L
    return _lib


class Config(C.Structure):
    """``gcn10_config`` of include/gcn10_host.h."""
    _fields_ = [("hysogs_data_path", C.c_char_p), ("esa_data_path", C.c_char_p),
                ("blocks_shp_path", C.c_char_p), ("lookup_table_path", C.c_char_p),
                ("log_dir", C.c_char_p), ("gpus", C.c_int), ("workers_per_gpu", C.c_int),
                ("strip_rows", C.c_int),
                ("io_threads", C.c_int), ("deflate_level", C.c_int), ("esa_tile_dir", C.c_char_p),
                ("gpu_deflate", C.c_int), ("gpu_inflate", C.c_int), ("direct_io", C.c_int),
                ("prefetch_blocks", C.c_int),
                ("table_mask", C.c_uint), ("cond_mask", C.c_uint)]


class Blocks(C.Structure):
    _fields_ = [("n", C.c_int), ("id", C.POINTER(C.c_int)), ("bbox", C.POINTER(C.c_double * 4))]


class HostError(RuntimeError):
    pass


def parse_config(path: str) -> dict:
    """src/config.c:44-114 -> dict of the keys; raises HostError like the reference aborts."""
    cfg = Config()
    err = C.create_string_buffer(1024)
    rc = lib().gcn10_config_parse(os.fsencode(path), C.byref(cfg), err, 1024)
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))
    out = {}
    for name, _t in Config._fields_:
        v = getattr(cfg, name)
        out[name] = v.decode() if isinstance(v, bytes) else v
    lib().gcn10_config_free(C.byref(cfg))
    return out


class Log:
    def __init__(self, log_dir: str, rank: int):
        self._h = lib().gcn10_log_open(os.fsencode(log_dir), rank)

    def message(self, level, msg, also_console=False):
        lib().gcn10_log_message(self._h, None if level is None else level.encode(),
                                None if msg is None else msg.encode(), also_console)

    def close(self):
        if self._h:
            lib().gcn10_log_close(self._h)
            self._h = None


def read_block_list(path: str):
    n = C.c_int(0)
    p = lib().gcn10_read_block_list(os.fsencode(path), C.byref(n))
    if not p:
        return None
    ids = [p[i] for i in range(n.value)]
    lib().free_(C.cast(p, C.c_void_p))
    return ids


def read_blocks_shapefile(path: str):
    """-> (ids list, bbox float64[n,4] {minx,miny,maxx,maxy})."""
    b = Blocks()
    err = C.create_string_buffer(1024)
    if lib().gcn10_blocks_open(os.fsencode(path), C.byref(b), err, 1024) != 0:
        raise HostError(err.value.decode(errors="replace"))
    ids = [b.id[i] for i in range(b.n)]
    bbox = np.array([list(b.bbox[i]) for i in range(b.n)], dtype=np.float64).reshape(b.n, 4)
    lib().gcn10_blocks_free(C.byref(b))
    return ids, bbox


class _ChunkRef(C.Structure):
    """``struct gcn10_chunk_ref`` of csrc/host/host_internal.h."""
    _fields_ = [("fd", C.c_int), ("file_off", C.c_uint64), ("nbytes", C.c_uint32), ("chunk_w", C.c_uint32),
                ("rows", C.c_uint32), ("src_x", C.c_uint32), ("src_y", C.c_uint32), ("copy_w", C.c_uint32),
                ("copy_h", C.c_uint32), ("dst_x", C.c_uint32), ("dst_y", C.c_uint32), ("flags", C.c_uint32),
                ("out_len", C.c_uint32)]


class _ReadPlan(C.Structure):
    """``struct gcn10_read_plan`` of csrc/host/host_internal.h."""
    _fields_ = [("chunks", C.POINTER(_ChunkRef)), ("n", C.c_size_t), ("cap", C.c_size_t),
                ("opened", C.c_void_p), ("n_opened", C.c_int), ("covered", C.c_uint64),
                ("max_chunk_bytes", C.c_uint32), ("staged_bytes", C.c_uint64)]


class Raster:
    """An open GeoTIFF / VRT (``gcn10_raster``)."""

    def __init__(self, path: str, tile_dir: str | None = None):
        err = C.create_string_buffer(1024)
        self._h = lib().gcn10_raster_open(os.fsencode(path),
                                          os.fsencode(tile_dir) if tile_dir else None, err, 1024)
        if not self._h:
            raise HostError(err.value.decode(errors="replace"))
        xs, ys = C.c_int(), C.c_int()
        gt = np.empty(6, dtype=np.float64)
        lib().gcn10_raster_info(self._h, C.byref(xs), C.byref(ys), gt)
        self.xsize, self.ysize, self.gt = xs.value, ys.value, gt.tolist()

    def read(self, xoff, yoff, xcount, ycount) -> np.ndarray:
        out = np.empty((ycount, xcount), dtype=np.uint8)
        err = C.create_string_buffer(1024)
        if lib().gcn10_raster_read(self._h, xoff, yoff, xcount, ycount, out.ctypes.data, err, 1024) != 0:
            raise HostError(err.value.decode(errors="replace"))
        return out

    def georef_ptr(self):
        return lib().gcn10_raster_georef(self._h)

    def plan(self, xoff, yoff, xcount, ycount):
        """The read plan the pipeline hands to the GPU decoder (gcn10_raster_plan_window,
        host_internal.h): None when the window has to go through the host reader, else
        (chunks, covered_pixels, max_chunk_bytes) with chunks = list of dicts holding the
        compressed bytes of a tile or strip and where its wanted part goes."""
        plan = _ReadPlan()
        err = C.create_string_buffer(1024)
        L = lib()
        L.gcn10_raster_plan_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(_ReadPlan), C.c_char_p, C.c_size_t]
        L.gcn10_raster_plan_window.restype = C.c_int
        L.gcn10_read_plan_free.argtypes = [C.POINTER(_ReadPlan)]
        L.gcn10_read_plan_free.restype = None
        rc = L.gcn10_raster_plan_window(self._h, xoff, yoff, xcount, ycount, C.byref(plan), err, 1024)
        if rc < 0:
            raise HostError(err.value.decode(errors="replace"))
        if rc > 0:
            return None
        try:
            chunks = []
            for i in range(plan.n):
                c = plan.chunks[i]
                chunks.append({"data": os.pread(c.fd, c.nbytes, c.file_off), "chunk_w": c.chunk_w, "rows": c.rows,
                               "src_x": c.src_x, "src_y": c.src_y, "copy_w": c.copy_w, "copy_h": c.copy_h,
                               "dst_x": c.dst_x, "dst_y": c.dst_y, "flags": c.flags, "out_len": c.out_len})
            return chunks, plan.covered, plan.max_chunk_bytes
        finally:
            L.gcn10_read_plan_free(C.byref(plan))

    def close(self):
        if self._h:
            lib().gcn10_raster_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def save_raster(data: np.ndarray, gt, path: str, georef_ptr=None, level: int = 0):
    """src/raster.c:192-227: tiled DEFLATE GeoTIFF of a uint8 raster."""
    a = np.ascontiguousarray(data, dtype=np.uint8)
    err = C.create_string_buffer(1024)
    rc = lib().gcn10_save_raster(a.ctypes.data, a.shape[1], a.shape[0], _f(gt, 6), georef_ptr,
                                 os.fsencode(path), level, err, 1024)
    if rc != 0:
        raise HostError(err.value.decode(errors="replace"))


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

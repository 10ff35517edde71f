"""ctypes face of ``libgcn10_gpu.so`` -- the C ABI of ``include/gcn10_gpu.h``.

Nothing here computes: every method forwards to the HIP library, and a missing
library or a missing gfx950 device raises (there is no CPU fallback, by design).
Device memory is owned through the ABI's own allocator, so neither this module
nor the tests need torch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import host as _host

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCN10_GPU_LIB") or os.path.join(_HERE, "libgcn10_gpu.so")
_lib = None

N_TABLES = 9
N_RASTERS = 18
COND_DRAINED = 1
COND_UNDRAINED = 2
ALL_TABLES = 0x1FF
NODATA = 255

#: every symbol include/gcn10_gpu.h declares (checked by the CPU test-suite)
ABI_SYMBOLS = (
    "gcn10_gpu_abi_version", "gcn10_gpu_device_count", "gcn10_gpu_init", "gcn10_gpu_destroy",
    "gcn10_gpu_last_error", "gcn10_gpu_device_info", "gcn10_gpu_malloc", "gcn10_gpu_free",
    "gcn10_gpu_host_alloc", "gcn10_gpu_host_free", "gcn10_gpu_memcpy_h2d",
    "gcn10_gpu_memcpy_d2h", "gcn10_gpu_memset", "gcn10_gpu_stream_create",
    "gcn10_gpu_stream_destroy", "gcn10_gpu_stream_sync", "gcn10_gpu_device_sync",
    "gcn10_gpu_event_create", "gcn10_gpu_event_destroy", "gcn10_gpu_event_record",
    "gcn10_gpu_event_sync", "gcn10_gpu_stream_wait_event", "gcn10_gpu_event_elapsed_ms",
    "gcn10_gpu_set_tables", "gcn10_gpu_resample", "gcn10_gpu_modify_hysogs_data",
    "gcn10_gpu_calculate_cn", "gcn10_gpu_prepare_tile", "gcn10_gpu_cn_strip",
    "gcn10_gpu_strip_algorithmic_bytes", "gcn10_gpu_last_kernel_name", "gcn10_gpu_set_option",
    "gcn10_gpu_deflate_arena_bound", "gcn10_gpu_deflate_strip", "gcn10_gpu_time_next_strip",
    "gcn10_gpu_pci_bus_id", "gcn10_gpu_deflate_fused_strip",
    "gcn10_gpu_deflate_fused_available",
    "gcn10_gpu_inflate_tiles", "gcn10_gpu_stream_copy", "gcn10_gpu_tune_single_raster",
    "gcn10_gpu_soil_words_state",
)


# struct gcn10_inflate_tile (include/gcn10_gpu.h)
TILE_RAW, TILE_PREDICTOR2 = 1, 2          # gcn10_inflate_tile.flags (include/gcn10_gpu.h)
INFLATE_TILE_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("chunk_w", "<u4"),
                               ("src_x", "<u4"), ("src_y", "<u4"), ("copy_w", "<u4"), ("copy_h", "<u4"),
                               ("flags", "<u4"), ("dst_off", "<u8")])


class Gcn10GpuError(RuntimeError):
    def __init__(self, code: int, where: str, message: str):
        super().__init__("%s failed (%d): %s" % (where, code, message))
        self.code = code


def lib():
    """Loads the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `make gpu` (or __graft_entry__.build()); "
                          "there is no CPU fallback for the CN path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i, sz, u = C.c_void_p, C.c_int, C.c_size_t, C.c_uint
        sig = {
            "gcn10_gpu_abi_version": (i, []),
            "gcn10_gpu_device_count": (i, []),
            "gcn10_gpu_init": (i, [i, C.POINTER(vp)]),
            "gcn10_gpu_destroy": (None, [vp]),
            "gcn10_gpu_last_error": (C.c_char_p, []),
            "gcn10_gpu_device_info": (i, [vp, C.c_char_p, sz, C.POINTER(sz)]),
            "gcn10_gpu_malloc": (i, [vp, sz, C.POINTER(vp)]),
            "gcn10_gpu_free": (i, [vp, vp]),
            "gcn10_gpu_host_alloc": (i, [vp, sz, C.POINTER(vp)]),
            "gcn10_gpu_host_free": (i, [vp, vp]),
            "gcn10_gpu_memcpy_h2d": (i, [vp, vp, vp, sz, vp]),
            "gcn10_gpu_memcpy_d2h": (i, [vp, vp, vp, sz, vp]),
            "gcn10_gpu_memset": (i, [vp, vp, i, sz, vp]),
            "gcn10_gpu_stream_create": (i, [vp, C.POINTER(vp)]),
            "gcn10_gpu_stream_destroy": (i, [vp, vp]),
            "gcn10_gpu_stream_sync": (i, [vp, vp]),
            "gcn10_gpu_device_sync": (i, [vp]),
            "gcn10_gpu_event_create": (i, [vp, C.POINTER(vp)]),
            "gcn10_gpu_event_destroy": (i, [vp, vp]),
            "gcn10_gpu_event_record": (i, [vp, vp, vp]),
            "gcn10_gpu_event_sync": (i, [vp, vp]),
            "gcn10_gpu_stream_wait_event": (i, [vp, vp, vp]),
            "gcn10_gpu_event_elapsed_ms": (i, [vp, vp, vp, C.POINTER(C.c_float)]),
            "gcn10_gpu_set_tables": (i, [vp, vp, i]),
            "gcn10_gpu_resample": (i, [vp, vp, i, i, vp, vp, i, i, vp, vp]),
            "gcn10_gpu_modify_hysogs_data": (i, [vp, vp, sz, i, vp]),
            "gcn10_gpu_calculate_cn": (i, [vp, vp, vp, sz, i, vp, vp]),
            "gcn10_gpu_prepare_tile": (i, [vp, vp, i, i, vp, i, vp]),
            "gcn10_gpu_cn_strip": (i, [vp, vp, i, i, vp, u, u, C.POINTER(vp), vp]),
            "gcn10_gpu_strip_algorithmic_bytes": (sz, [i, i, i, i, u, u]),
            "gcn10_gpu_last_kernel_name": (C.c_char_p, [vp]),
            "gcn10_gpu_set_option": (i, [vp, C.c_char_p, i]),
            "gcn10_gpu_time_next_strip": (i, [vp, vp, vp]),
            "gcn10_gpu_stream_copy": (i, [vp, vp, vp, sz, vp]),
            "gcn10_gpu_tune_single_raster": (i, [vp, vp, i, i, vp, u, u, vp, sz, sz, C.POINTER(vp),
                                                 C.POINTER(C.c_float), C.c_char_p, sz, vp]),
            "gcn10_gpu_pci_bus_id": (i, [i, C.c_char_p, sz]),
            "gcn10_gpu_soil_words_state": (i, [vp, vp]),
            "gcn10_gpu_deflate_fused_strip": (i, [vp, vp, i, i, vp, u, u, vp, sz, vp, vp, vp]),
            "gcn10_gpu_deflate_fused_available": (i, [vp]),
            "gcn10_gpu_inflate_tiles": (i, [vp, vp, vp, i, u, vp, sz, vp, vp]),
            "gcn10_gpu_deflate_arena_bound": (sz, [i, i, i]),
            "gcn10_gpu_deflate_strip": (i, [vp, vp, i, i, i, vp, sz, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count() -> int:
    return int(lib().gcn10_gpu_device_count())


def strip_algorithmic_bytes(W: int, rows: int, hsx: int, hsy: int, cond_mask: int,
                            table_mask: int) -> int:
    return int(lib().gcn10_gpu_strip_algorithmic_bytes(W, rows, hsx, hsy, cond_mask, table_mask))


class DevBuf:
    """A device allocation made through the C ABI (freed with the engine or on close())."""

    def __init__(self, eng: "Engine", nbytes: int):
        self.eng = eng
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        eng._chk(lib().gcn10_gpu_malloc(eng._ctx, self.nbytes, C.byref(p)), "gcn10_gpu_malloc")
        self.ptr = p.value or 0
        eng._bufs.append(self)

    def at(self, offset: int) -> int:
        assert 0 <= offset <= self.nbytes
        return self.ptr + offset

    def close(self):
        if self.ptr:
            lib().gcn10_gpu_free(self.eng._ctx, self.ptr)
            self.ptr = 0
            if self in self.eng._bufs:
                self.eng._bufs.remove(self)


class Engine:
    """One GPU context (``gcn10_gpu_ctx``): the device-side stand-in of one MPI rank."""

    def __init__(self, device: int = 0):
        self._ctx = C.c_void_p()
        self._bufs = []
        rc = lib().gcn10_gpu_init(device, C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise Gcn10GpuError(rc, "gcn10_gpu_init", lib().gcn10_gpu_last_error().decode())
        self.device = device
        self.n_tables = 0

    # -- plumbing ---------------------------------------------------------
    def _chk(self, rc: int, where: str):
        if rc != 0:
            raise Gcn10GpuError(rc, where, lib().gcn10_gpu_last_error().decode())

    def close(self):
        if self._ctx:
            for b in list(self._bufs):
                b.close()
            lib().gcn10_gpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        hbm = C.c_size_t()
        cus = lib().gcn10_gpu_device_info(self._ctx, name, 256, C.byref(hbm))
        if cus < 0:
            self._chk(cus, "gcn10_gpu_device_info")
        return {"name": name.value.decode(), "cus": cus, "hbm_bytes": hbm.value}

    def pci_bus_id(self) -> str:
        """PCI bus id of this engine's device, e.g. 0000:05:00.0."""
        buf = C.create_string_buffer(32)
        self._chk(lib().gcn10_gpu_pci_bus_id(self.device, buf, 32), "gcn10_gpu_pci_bus_id")
        return buf.value.decode()

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def upload(self, arr: np.ndarray, stream=None) -> DevBuf:
        a = np.ascontiguousarray(arr)
        buf = DevBuf(self, max(a.nbytes, 1))
        self.h2d(buf.ptr, a, stream)
        self.sync(stream)
        return buf

    def h2d(self, dptr: int, arr: np.ndarray, stream=None):
        a = np.ascontiguousarray(arr)
        self._chk(lib().gcn10_gpu_memcpy_h2d(self._ctx, dptr, a.ctypes.data, a.nbytes, stream),
                  "gcn10_gpu_memcpy_h2d")

    def download(self, dptr: int, shape, dtype=np.uint8, stream=None) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            self._chk(lib().gcn10_gpu_memcpy_d2h(self._ctx, out.ctypes.data, dptr, out.nbytes,
                                                 stream), "gcn10_gpu_memcpy_d2h")
        self.sync(stream)
        return out

    def memset(self, dptr: int, value: int, nbytes: int, stream=None):
        self._chk(lib().gcn10_gpu_memset(self._ctx, dptr, value, nbytes, stream), "gcn10_gpu_memset")

    def sync(self, stream=None):
        self._chk(lib().gcn10_gpu_stream_sync(self._ctx, stream), "gcn10_gpu_stream_sync")

    def device_sync(self):
        self._chk(lib().gcn10_gpu_device_sync(self._ctx), "gcn10_gpu_device_sync")

    def stream_create(self) -> int:
        s = C.c_void_p()
        self._chk(lib().gcn10_gpu_stream_create(self._ctx, C.byref(s)), "gcn10_gpu_stream_create")
        return s.value

    def stream_destroy(self, s):
        self._chk(lib().gcn10_gpu_stream_destroy(self._ctx, s), "gcn10_gpu_stream_destroy")

    def event_create(self) -> int:
        e = C.c_void_p()
        self._chk(lib().gcn10_gpu_event_create(self._ctx, C.byref(e)), "gcn10_gpu_event_create")
        return e.value

    def event_destroy(self, e):
        self._chk(lib().gcn10_gpu_event_destroy(self._ctx, e), "gcn10_gpu_event_destroy")

    def event_record(self, e, stream=None):
        self._chk(lib().gcn10_gpu_event_record(self._ctx, e, stream), "gcn10_gpu_event_record")

    def event_sync(self, e):
        self._chk(lib().gcn10_gpu_event_sync(self._ctx, e), "gcn10_gpu_event_sync")

    def stream_wait_event(self, stream, e):
        self._chk(lib().gcn10_gpu_stream_wait_event(self._ctx, stream, e),
                  "gcn10_gpu_stream_wait_event")

    def elapsed_ms(self, e0, e1) -> float:
        ms = C.c_float()
        self._chk(lib().gcn10_gpu_event_elapsed_ms(self._ctx, e0, e1, C.byref(ms)),
                  "gcn10_gpu_event_elapsed_ms")
        return float(ms.value)

    def set_option(self, name: str, value: int):
        self._chk(lib().gcn10_gpu_set_option(self._ctx, name.encode(), int(value)),
                  "gcn10_gpu_set_option")

    def time_next_strip(self, e0, e1):
        self._chk(lib().gcn10_gpu_time_next_strip(self._ctx, e0, e1), "gcn10_gpu_time_next_strip")

    def tune_single_raster(self, esa, W: int, rows: int, cj, cond_mask: int, table_mask: int, arena, arena_bytes: int,
                           step: int = 16 << 20, stream=None):
        """gcn10_gpu_tune_single_raster: (best raster pointer inside the arena, best ms, report dict)."""
        import json as _json
        best = C.c_void_p()
        ms = C.c_float()
        rep = C.create_string_buffer(512)
        self._chk(lib().gcn10_gpu_tune_single_raster(self._ctx, esa, W, rows, cj, cond_mask, table_mask, arena,
                                                     int(arena_bytes), int(step), C.byref(best), C.byref(ms), rep, 512,
                                                     stream), "gcn10_gpu_tune_single_raster")
        return best.value, float(ms.value), _json.loads(rep.value.decode() or "{}")

    def stream_copy(self, src, dst, nbytes: int, stream=None):
        """Plain 1R:1W copy with the strip kernel's launch shape (the same-run streaming ceiling)."""
        self._chk(lib().gcn10_gpu_stream_copy(self._ctx, src, dst, int(nbytes), stream), "gcn10_gpu_stream_copy")

    def soil_words_state(self, stream=None) -> int:
        """0 = code bytes (option off), 1 = compact words, 2 = code bytes (a column group of the prepared tile
        spans more than two soil cells); see include/gcn10_gpu.h."""
        rc = lib().gcn10_gpu_soil_words_state(self._ctx, stream)
        if rc < 0:
            self._chk(rc, "gcn10_gpu_soil_words_state")
        return rc

    def last_kernel_name(self) -> str:
        return lib().gcn10_gpu_last_kernel_name(self._ctx).decode()

    # -- the reference's functions ------------------------------------------
    def set_tables(self, tables: np.ndarray):
        """int32[n,256,5] reference-format tables (src/cn.c:148)."""
        t = np.ascontiguousarray(tables, dtype=np.int32)
        if t.ndim == 2:
            t = t[None]
        if t.ndim != 3 or t.shape[1:] != (256, 5):
            raise ValueError("tables must be int32[n,256,5]")
        self._chk(lib().gcn10_gpu_set_tables(self._ctx, t.ctypes.data, t.shape[0]),
                  "gcn10_gpu_set_tables")
        self.n_tables = t.shape[0]

    def resample(self, coarse_d: int, hsx: int, hsy: int, ci_d: int, cj_d: int, W: int,
                 rows: int, out_d: int, stream=None):
        self._chk(lib().gcn10_gpu_resample(self._ctx, coarse_d, hsx, hsy, ci_d, cj_d, W, rows,
                                           out_d, stream), "gcn10_gpu_resample")

    def modify_hysogs_data(self, h_d: int, npix: int, drained: bool, stream=None):
        self._chk(lib().gcn10_gpu_modify_hysogs_data(self._ctx, h_d, npix, int(bool(drained)),
                                                     stream), "gcn10_gpu_modify_hysogs_data")

    def calculate_cn(self, esa_d: int, hsg_d: int, npix: int, table_index: int, out_d: int,
                     stream=None):
        self._chk(lib().gcn10_gpu_calculate_cn(self._ctx, esa_d, hsg_d, npix, table_index, out_d,
                                               stream), "gcn10_gpu_calculate_cn")

    def prepare_tile(self, coarse_d: int, hsx: int, hsy: int, ci_d: int, W: int, stream=None):
        self._chk(lib().gcn10_gpu_prepare_tile(self._ctx, coarse_d, hsx, hsy, ci_d, W, stream),
                  "gcn10_gpu_prepare_tile")

    def cn_strip(self, esa_d: int, W: int, rows: int, cj_d: int, cond_mask: int, table_mask: int,
                 outs: Sequence[Optional[int]], stream=None):
        arr = (C.c_void_p * N_RASTERS)()
        for r in range(N_RASTERS):
            arr[r] = outs[r] if r < len(outs) and outs[r] else None
        self._chk(lib().gcn10_gpu_cn_strip(self._ctx, esa_d, W, rows, cj_d, cond_mask, table_mask,
                                           arr, stream), "gcn10_gpu_cn_strip")

    # -- output encode (src/raster.c:204-219 on the GPU) ----------------------
    def deflate_rasters(self, raster_ptrs: Sequence[int], W: int, rows: int, stream=None):
        """zlib-encodes every 256x256 tile of the given device rasters.

        Returns (arena bytes as uint8 array, table uint32[n, down, across, 2])."""
        n = len(raster_ptrs)
        across, down = (W + 255) // 256, (rows + 255) // 256
        cap = int(lib().gcn10_gpu_deflate_arena_bound(W, rows, n))
        ptrs = self.upload(np.array(raster_ptrs, dtype=np.uint64))
        arena = self.alloc(cap)
        table = self.alloc(n * across * down * 8)
        cursor = self.alloc(8)
        try:
            self._chk(lib().gcn10_gpu_deflate_strip(self._ctx, ptrs.ptr, n, W, rows, arena.ptr, cap,
                                                    table.ptr, cursor.ptr, stream),
                      "gcn10_gpu_deflate_strip")
            used = int(self.download(cursor.ptr, (1,), dtype=np.uint64, stream=stream)[0])
            tab = self.download(table.ptr, (n, down, across, 2), dtype=np.uint32, stream=stream)
            data = self.download(arena.ptr, (min(used, cap),), stream=stream)
        finally:
            for b in (ptrs, arena, table, cursor):
                b.close()
        return data, tab, used

    def deflate_fused(self, esa_d: int, W: int, rows: int, cj_d: int, cond_mask: int = 3,
                      table_mask: int = ALL_TABLES, stream=None, arena_cap: Optional[int] = None):
        """Encoded tiles of the selected rasters straight from landcover + prepared soil
        (gcn10_gpu_deflate_fused_strip).  Returns (arena bytes, table uint32[n, down, across, 2], used).
        arena_cap: an arena smaller than gcn10_gpu_deflate_arena_bound (tests: streams that do not fit)."""
        n = bin(cond_mask & 3).count("1") * bin(table_mask & ALL_TABLES).count("1")
        across, down = (W + 255) // 256, (rows + 255) // 256
        cap = int(lib().gcn10_gpu_deflate_arena_bound(W, rows, n)) if arena_cap is None else int(arena_cap)
        arena = self.alloc(cap)
        table = self.alloc(n * across * down * 8)
        cursor = self.alloc(8)
        try:
            self._chk(lib().gcn10_gpu_deflate_fused_strip(self._ctx, esa_d, W, rows, cj_d, cond_mask,
                                                          table_mask, arena.ptr, cap, table.ptr, cursor.ptr,
                                                          stream), "gcn10_gpu_deflate_fused_strip")
            used = int(self.download(cursor.ptr, (1,), dtype=np.uint64, stream=stream)[0])
            tab = self.download(table.ptr, (n, down, across, 2), dtype=np.uint32, stream=stream)
            data = self.download(arena.ptr, (min(used, cap),), stream=stream)
        finally:
            for b in (arena, table, cursor):
                b.close()
        return data, tab, used

    def inflate_tiles(self, streams: Sequence[bytes], chunk_w: int, chunk_rows: Sequence[int],
                      windows: Sequence[tuple], dst_shape: tuple, stream=None, flags: Optional[Sequence[int]] = None,
                      out_lens: Optional[Sequence[int]] = None):
        """Decodes zlib streams on the GPU (gcn10_gpu_inflate_tiles).

        streams[i] decodes to a chunk of chunk_rows[i] x chunk_w pixels; windows[i] =
        (src_x, src_y, copy_w, copy_h, dst_x, dst_y) places part of it in a zero-filled uint8
        raster of dst_shape.  flags[i]: TILE_RAW (streams[i] is the chunk's pixels as they are) |
        TILE_PREDICTOR2 (rows are horizontal differences); out_lens[i] overrides the decoded size
        (a raw chunk staged from its first wanted row on).  Returns (raster, status uint32[n])."""
        n = len(streams)
        H, W = dst_shape
        tiles = np.zeros(n, dtype=INFLATE_TILE_DTYPE)
        parts, off = [], 0
        for k, st in enumerate(streams):
            sx, sy, cw, ch, dx, dy = windows[k]
            tiles[k] = (off, len(st), out_lens[k] if out_lens else chunk_w * chunk_rows[k], chunk_w, sx, sy, cw, ch,
                        flags[k] if flags else 0, dy * W + dx)
            pad = (-len(st)) % 16 + 16
            parts.append(st)
            parts.append(bytes(pad))
            off += len(st) + pad
        comp = np.frombuffer(b"".join(parts) or bytes(16), dtype=np.uint8)
        bufs = [self.upload(comp, stream), self.upload(tiles.view(np.uint8), stream), self.alloc(max(H * W, 1)),
                self.alloc(4 * max(n, 1))]
        try:
            self.memset(bufs[2].ptr, 0, max(H * W, 1), stream)
            self.memset(bufs[3].ptr, 0xFF, 4 * max(n, 1), stream)
            self._chk(lib().gcn10_gpu_inflate_tiles(self._ctx, bufs[0].ptr, bufs[1].ptr, n,
                                                    max(chunk_w * max(chunk_rows, default=1), 1), bufs[2].ptr, W,
                                                    bufs[3].ptr, stream), "gcn10_gpu_inflate_tiles")
            out = self.download(bufs[2].ptr, (H, W), stream=stream)
            status = self.download(bufs[3].ptr, (n,), dtype=np.uint32, stream=stream)
        finally:
            for b in bufs:
                b.close()
        return out, status

    # -- block level: src/cn.c:205-290 in memory -----------------------------
    def process_block_mem(self, esa: np.ndarray, gt, coarse: np.ndarray, soil_gt,
                          cond_mask: int = 3, table_mask: int = ALL_TABLES,
                          strip_rows: Optional[int] = None) -> np.ndarray:
        """What process_block() computes between load_raster and save_raster.

        Returns uint8[18,H,W] in the reference's raster order (unselected
        rasters stay 0).  Index maps come from the host library, every pixel
        from the HIP kernels.
        """
        esa = np.ascontiguousarray(esa, dtype=np.uint8)
        coarse = np.ascontiguousarray(coarse, dtype=np.uint8)
        H, W = esa.shape
        hsy, hsx = coarse.shape
        out = np.zeros((N_RASTERS, H, W), dtype=np.uint8)
        if H == 0 or W == 0:
            return out
        ci, cj = _host.build_index_maps(gt, soil_gt, W, H, hsx, hsy)
        bufs = []
        try:
            esa_d = self.upload(esa); bufs.append(esa_d)
            coarse_d = self.upload(coarse); bufs.append(coarse_d)
            ci_d = self.upload(ci); bufs.append(ci_d)
            cj_d = self.upload(cj); bufs.append(cj_d)
            sel = [r for r in range(N_RASTERS)
                   if (cond_mask >> (r // 9)) & 1 and (table_mask >> (r % 9)) & 1]
            out_d = {}
            for r in sel:
                out_d[r] = self.alloc(H * W)
                bufs.append(out_d[r])
                self.memset(out_d[r].ptr, 0xA5, H * W)     # poison: every byte must be written
            self.prepare_tile(coarse_d.ptr, hsx, hsy, ci_d.ptr, W)
            step = H if not strip_rows else int(strip_rows)
            for y0 in range(0, H, step):
                rows = min(step, H - y0)
                ptrs = [out_d[r].at(y0 * W) if r in out_d else None for r in range(N_RASTERS)]
                self.cn_strip(esa_d.at(y0 * W), W, rows, cj_d.at(4 * y0), cond_mask, table_mask,
                              ptrs)
            self.sync()
            for r in sel:
                out[r] = self.download(out_d[r].ptr, (H, W))
        finally:
            for b in bufs:
                b.close()
        return out

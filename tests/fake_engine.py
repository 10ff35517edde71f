"""A stand-in for gcn10_amd.gpu.Engine that touches no GPU: TEST INFRASTRUCTURE ONLY.

bench.py loads it only when GCN10_BENCH_ENGINE=tests.fake_engine:FakeEngine is set (the launcher /
aggregation tests, which run where there is no GPU); the bench line then carries
"data": "FAKE ENGINE ...".  It computes nothing: every launch sleeps, events carry host clocks.
Never shipped as a fallback -- gcn10_amd never imports it.
"""
import os
import time

import numpy as np


class _Buf:
    _count = 0

    def __init__(self, n):
        _Buf._count += 1
        self.serial = _Buf._count
        self.ptr = 0x100000000 * self.serial        # distinct, far apart: the stand-in never dereferences them
        self.n = n
        self.closed = False

    def at(self, off):
        return self.ptr + off

    def close(self):
        self.closed = True


class FakeEngine:
    LAUNCH_S = float(os.environ.get("GCN10_FAKE_LAUNCH_S", "0.002"))

    @staticmethod
    def device_count():
        return int(os.environ.get("GCN10_FAKE_DEVICES", "2"))

    def __init__(self, device):
        if not (0 <= device < self.device_count()):
            raise RuntimeError("fake device %d out of range" % device)
        self.device = device
        self._pending = None
        self._last = "fake_kernel"
        # rank-dependent speed: rank 1 is the slow one (the aggregation must take the max)
        self._scale = 1.0 + 0.5 * device

    def device_info(self):
        return {"name": "fake device %d" % self.device, "cus": 1, "hbm_bytes": 0}

    def pci_bus_id(self):
        return "0000:%02x:00.0" % (0x10 + self.device)

    def set_tables(self, tables):
        pass

    def upload(self, a):
        return _Buf(int(np.asarray(a).nbytes))

    def alloc(self, n):
        return _Buf(int(n))

    def resample(self, *a):
        pass

    def set_option(self, name, value):
        pass

    def tune_single_raster(self, esa, W, rows, cj, cond_mask, table_mask, arena, arena_bytes, step, stream=None):
        """The calibration entry of the real engine, without launches: a time that depends on the arena only
        (every fifth allocation is the 'fast' one), the best position a step into it."""
        serial = arena // 0x100000000
        ms = 2.0 * self._scale * (0.9 if serial % 5 == 0 else 1.0) + 1e-4 * (serial % 7)
        positions = max(1, (arena_bytes - W * rows) // step + 1)
        report = {"positions": int(positions), "step_bytes": int(step), "shapes": 8, "best_ms": round(ms, 4),
                  "worst_ms": round(ms * 1.1, 4), "best_offset_bytes": int(step), "xcd_slabs": 1,
                  "grid_blocks_per_cu": 8, "ilp1": 2, "prefetch": 1}
        return arena + step, ms, report

    def calculate_cn(self, *a):
        self._launch()

    def prepare_tile(self, *a):
        pass

    def event_create(self):
        return [0.0]

    def event_record(self, ev):
        ev[0] = time.perf_counter()

    def time_next_strip(self, e0, e1):
        self._pending = (e0, e1)

    def _launch(self):
        t0 = time.perf_counter()
        time.sleep(self.LAUNCH_S * self._scale)
        if self._pending:
            self._pending[0][0], self._pending[1][0] = t0, time.perf_counter()
            self._pending = None

    def cn_strip(self, *a):
        self._last = "fake_strip_kernel"
        self._launch()

    def stream_copy(self, *a):
        self._launch()

    def elapsed_ms(self, e0, e1):
        return (e1[0] - e0[0]) * 1e3

    def last_kernel_name(self):
        return self._last

    def sync(self, stream=None):
        pass

    def device_sync(self):
        pass

    def close(self):
        pass

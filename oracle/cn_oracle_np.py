"""numpy restatement of gcn10's curve-number hot path (second, independent oracle).

TEST INFRASTRUCTURE ONLY -- imported by tests/, ``__graft_entry__.smoke()`` and
nothing else.  PARITY UNPINNED: like ``cn_oracle.c`` it is a restatement of the
reference source (file:line cited per function), pinned only by the shipped
lookup CSVs; the reference holds no golden rasters and cannot be built here
(``src/global.h:8-13`` needs ``gdal.h``).  Its job is to catch slips in the C
oracle: the two are written separately (vectorised here, loop-shaped there) and
tests require them to agree bit for bit.
"""
from __future__ import annotations

import numpy as np

NODATA = 255
CONDS = ("drained", "undrained")          # src/cn.c:145
HCS = ("p", "f", "g")                     # src/cn.c:146
ARCS = ("i", "ii", "iii")                 # src/cn.c:147


def _atoi(tok: bytes) -> int:
    """C ``atoi``: optional whitespace, optional sign, leading digits, else 0."""
    i, n = 0, len(tok)
    while i < n and tok[i:i + 1] in b" \t\n\v\f\r":
        i += 1
    sign = 1
    if i < n and tok[i:i + 1] in b"+-":
        sign = -1 if tok[i:i + 1] == b"-" else 1
        i += 1
    v = 0
    while i < n and 48 <= tok[i] <= 57:
        v = v * 10 + (tok[i] - 48)
        i += 1
    return sign * v


def _fgets_lines(data: bytes, bufsize: int):
    """Yield what successive ``fgets(buf, bufsize, f)`` calls would return."""
    pos, n = 0, len(data)
    while pos < n:
        end = data.find(b"\n", pos, pos + bufsize - 1)
        stop = end + 1 if end >= 0 else min(n, pos + bufsize - 1)
        yield data[pos:stop]
        pos = stop


def _strtok_fields(line: bytes):
    """Non-empty comma-separated fields, as ``strtok(.., ",")`` walks them."""
    return [f for f in line.split(b",") if f]


def load_lookup_table(path: str):
    """src/cn.c:13-85.  Returns (table int32[256,5], n_bad_rows)."""
    with open(path, "rb") as f:
        data = f.read()
    # a NUL byte would end the C string early; the shipped files have none
    table = np.full((256, 5), NODATA, dtype=np.int32)         # :36-40
    lines = _fgets_lines(data, 128)                           # char line[128], :17
    try:
        next(lines)                                           # header, :43
    except StopIteration:
        raise ValueError("empty lookup table") from None      # :44-47
    bad = 0
    for line in lines:                                        # :51
        fields = _strtok_fields(line)                         # :52
        if not fields:
            continue                                          # :53-55
        code = fields[0]
        us = code.find(b"_")                                  # :57
        if us < 0:
            bad += 1                                          # :58-63
            continue
        lc = _atoi(code[:us])                                 # :65
        letter = code[us + 1:us + 2]
        sg = {b"A": 1, b"B": 2, b"C": 3}.get(letter, 4)       # :66
        if len(fields) < 2:
            bad += 1                                          # :68-73
            continue
        cn = _atoi(fields[1])                                 # :74
        if 0 <= lc < 256:                                     # :75-77
            table[lc, sg] = cn
        else:
            bad += 1                                          # :78-82
    return table, bad


def modify_hysogs_data(h: np.ndarray, drained: bool) -> np.ndarray:
    """src/cn.c:88-111 (returns a new array)."""
    h = h.copy()
    dual = (h >= 11) & (h <= 14)
    if drained:
        h[dual] = 4                                           # :92-98
    else:
        h[dual] = h[dual] - 10                                # :99-110
    return h


def calculate_cn(esa: np.ndarray, hsg: np.ndarray, table: np.ndarray) -> np.ndarray:
    """src/cn.c:114-131 applied to a raster pre-filled with 255 (src/cn.c:289)."""
    out = np.full(esa.shape, NODATA, dtype=np.uint8)
    ok = hsg < 5                                              # :123-124
    v = table[esa[ok].astype(np.intp), hsg[ok].astype(np.intp)]   # :125
    hit = v < NODATA                                          # :126
    res = out[ok]
    res[hit] = (v[hit] & 0xFF).astype(np.uint8)               # (uint8_t) cast, :127
    out[ok] = res
    return out


def _c_round(v: np.ndarray) -> np.ndarray:
    """C99 ``round``: nearest, ties away from zero (numpy's round is ties-even)."""
    with np.errstate(invalid="ignore"):
        t = np.trunc(v)
        frac = np.abs(v - t)        # exact for doubles; inf - inf = nan stays put
        return t + np.where(frac >= 0.5, np.copysign(1.0, v), 0.0)


def _to_int_x86(v: np.ndarray) -> np.ndarray:
    """(int)double as cvttsd2si does it: NaN / out of range -> INT_MIN."""
    ok = (v > -2147483649.0) & (v < 2147483648.0)
    out = np.full(v.shape, -2**31, dtype=np.int64)
    out[ok] = np.trunc(v[ok]).astype(np.int64)
    return out


def index_maps(gt, soil_gt, esax: int, esay: int, hsx: int, hsy: int):
    """The separable form of src/cn.c:218-229: ci[x], cj[y] (int32)."""
    gt = np.asarray(gt, dtype=np.float64)
    sg = np.asarray(soil_gt, dtype=np.float64)
    x = np.arange(esax, dtype=np.float64)
    y = np.arange(esay, dtype=np.float64)
    px = gt[0] + (x + 0.5) * gt[1]                            # :222
    py = gt[3] + (y + 0.5) * gt[5]                            # :219
    with np.errstate(all="ignore"):
        dc = (px - sg[0]) / sg[1]                             # :223
        dr = (sg[3] - py) / np.abs(sg[5])                     # :224
    ci = np.clip(_to_int_x86(_c_round(dc)), 0, hsx - 1)       # :225,228
    cj = np.clip(_to_int_x86(_c_round(dr)), 0, hsy - 1)       # :226,229
    return ci.astype(np.int32), cj.astype(np.int32)


def resample(coarse: np.ndarray, gt, soil_gt, esax: int, esay: int) -> np.ndarray:
    """src/cn.c:218-232."""
    hsy, hsx = coarse.shape
    ci, cj = index_maps(gt, soil_gt, esax, esay, hsx, hsy)
    return coarse[cj[:, None], ci[None, :]]                   # :230


def window(t, rx: int, ry: int, bbox):
    """src/raster.c:126-162.  Returns (xoff, yoff, xcount, ycount, gt) or None."""
    t = [float(v) for v in t]
    minx, miny, maxx, maxy = (float(v) for v in bbox)
    f = lambda v: int(_to_int_x86(np.array([v], dtype=np.float64))[0])
    with np.errstate(all="ignore"):
        xo = f(np.floor(np.float64(minx - t[0]) / np.float64(t[1])))    # :127
        yo = f(np.floor(np.float64(maxy - t[3]) / np.float64(t[5])))    # :128
        xc = f(np.ceil(np.float64(maxx - minx) / np.float64(t[1])))     # :129
        yc = f(np.ceil(np.float64(miny - maxy) / np.float64(t[5])))     # :130
    if xo < 0:
        xc += xo
        xo = 0
    if yo < 0:
        yc += yo
        yo = 0
    if xo >= rx or yo >= ry or xc <= 0 or yc <= 0:            # :142-147
        return None
    xc = min(xc, rx - xo)                                     # :148-153
    yc = min(yc, ry - yo)
    gt = [t[0] + xo * t[1], t[1], t[2], t[3] + yo * t[5], t[4], t[5]]   # :157-162
    return xo, yo, xc, yc, gt


def process_block_mem(esa, gt, coarse, soil_gt, tables):
    """src/cn.c:205-380 without I/O.  tables: int32[9,256,5] in hc-major order.
    Returns uint8[18, H, W] ordered (cond, hc, arc) like src/cn.c:236-259."""
    esay, esax = esa.shape
    fine = resample(coarse, gt, soil_gt, esax, esay)
    out = np.empty((18, esay, esax), dtype=np.uint8)
    for c, cond in enumerate(CONDS):
        adj = modify_hysogs_data(fine, cond == "drained")
        for k in range(9):
            out[c * 9 + k] = calculate_cn(esa, adj, tables[k])
    return out


def fused_semantics(esa, fine_hsg, tables):
    """SURVEY.md section 8(a) 'fused semantics' written as table algebra --
    a third formulation used to cross-check process_block_mem."""
    t8 = np.where(tables < NODATA, tables & 0xFF, NODATA).astype(np.uint8)  # [9,256,5]
    h = np.arange(256)
    dual = (h >= 11) & (h <= 14)
    map_d = np.where(dual, 4, h)
    map_u = np.where(dual, h - 10, h)
    out = np.empty((18,) + esa.shape, dtype=np.uint8)
    for c, m in enumerate((map_d, map_u)):
        s = m[fine_hsg]
        ok = s < 5
        sc = np.where(ok, s, 0)
        for k in range(9):
            out[c * 9 + k] = np.where(ok, t8[k][esa, sc], NODATA)
    return out

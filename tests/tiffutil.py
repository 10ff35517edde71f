"""A small independent TIFF *producer* for reader tests (strips/tiles, none/deflate/LZW/
packbits, predictor 2, big-endian, BigTIFF) and a shapefile producer."""
import struct
import os
import zlib

import numpy as np


def lzw_encode(data: bytes) -> bytes:
    """TIFF 6.0 LZW (MSB first, early change), straightforward dictionary coder."""
    CLEAR, EOI = 256, 257
    out = bytearray()
    acc = 0
    nbits = 0
    width = 9

    def put(code):
        nonlocal acc, nbits
        acc = (acc << width) | code
        nbits += width
        while nbits >= 8:
            out.append((acc >> (nbits - 8)) & 0xFF)
            nbits -= 8
        acc &= (1 << nbits) - 1

    table = {bytes([i]): i for i in range(256)}
    nxt = 258
    put(CLEAR)
    w = b""
    for b in data:
        wb = w + bytes([b])
        if wb in table:
            w = wb
            continue
        put(table[w])
        table[wb] = nxt
        nxt += 1
        # the encoder's table runs one entry ahead of the decoder's, so it widens the
        # codes one entry later than the decoder's "early change" test (libtiff: free_ent > maxcode)
        if nxt == 4093:
            put(CLEAR)
            table = {bytes([i]): i for i in range(256)}
            nxt = 258
            width = 9
        elif nxt > (1 << width) - 1:
            width += 1
        w = bytes([b])
    if w:
        put(table[w])
        nxt += 1
        if nxt == 4093:
            put(CLEAR)
            width = 9
        elif nxt > (1 << width) - 1:
            width += 1
    put(EOI)
    if nbits:
        out.append((acc << (8 - nbits)) & 0xFF)
    return bytes(out)


def packbits_encode(data: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 127:
            j += 1
        run = j - i + 1
        if run >= 2:
            out += bytes([(257 - run) & 0xFF, data[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128 and not (j + 1 < n and data[j] == data[j + 1]):
            j += 1
        if j == i:
            j = i + 1
        out += bytes([j - i - 1]) + data[i:j]
        i = j
    return bytes(out)


def write_tiff(path, img, gt=None, compression=1, tile=None, rows_per_strip=None, predictor=1,
               big_endian=False, bigtiff=False, geokeys=None, pixel_is_point=False):
    """img: uint8[H,W].  tile=(tw,th) for tiles else strips.  Returns nothing."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    E = ">" if big_endian else "<"
    chunks = []
    if tile:
        tw, th = tile
        for ty in range(0, H, th):
            for tx in range(0, W, tw):
                c = np.zeros((th, tw), np.uint8)
                part = img[ty:ty + th, tx:tx + tw]
                c[:part.shape[0], :part.shape[1]] = part
                chunks.append(c)
    else:
        rps = rows_per_strip or H
        for y in range(0, H, rps):
            chunks.append(img[y:y + rps].copy())

    def enc(c):
        if predictor == 2:
            c = c.astype(np.int16)
            c[:, 1:] = c[:, 1:] - c[:, :-1]
            c = (c & 0xFF).astype(np.uint8)
        raw = c.tobytes()
        if compression == 1:
            return raw
        if compression in (8, 32946):
            return zlib.compress(raw, 6)
        if compression == 5:
            return lzw_encode(raw)
        if compression == 32773:
            return b"".join(packbits_encode(bytes(r)) for r in c)
        raise ValueError(compression)

    if compression in (8, 32946) and len(chunks) > 64:
        # big synthetic worlds (tools/bench_pipeline.py): zlib releases the GIL
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
            blobs = list(ex.map(enc, chunks))
    else:
        blobs = [enc(c) for c in chunks]
    off_sz = 8 if bigtiff else 4
    hdr_len = 16 if bigtiff else 8
    pos = hdr_len
    offsets = []
    body = bytearray()
    for b in blobs:
        offsets.append(pos)
        body += b
        pos += len(b)
        if pos & 1:
            body += b"\0"
            pos += 1
    counts = [len(b) for b in blobs]

    SHORT, LONG, DOUBLE, LONG8, ASCII = 3, 4, 12, 16, 2
    ents = [(256, LONG, [W]), (257, LONG, [H]), (258, SHORT, [8]), (259, SHORT, [compression]),
            (262, SHORT, [1]), (277, SHORT, [1]), (284, SHORT, [1])]
    offt = LONG8 if bigtiff else LONG
    if tile:
        ents += [(322, SHORT, [tile[0]]), (323, SHORT, [tile[1]]), (324, offt, offsets), (325, offt, counts)]
    else:
        ents += [(273, offt, offsets), (278, LONG, [rows_per_strip or H]), (279, offt, counts)]
    if predictor != 1:
        ents.append((317, SHORT, [predictor]))
    if gt is not None:
        ents.append((33550, DOUBLE, [gt[1], -gt[5], 0.0]))
        ents.append((33922, DOUBLE, [0.0, 0.0, 0.0, gt[0], gt[3], 0.0]))
    if geokeys is None:
        geokeys = [1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 2 if pixel_is_point else 1, 2048, 0, 1, 4326]
    ents.append((34735, SHORT, list(geokeys)))
    ents.sort(key=lambda e: e[0])
    fmt = {SHORT: "H", LONG: "I", DOUBLE: "d", LONG8: "Q", ASCII: "c"}
    size = {SHORT: 2, LONG: 4, DOUBLE: 8, LONG8: 8, ASCII: 1}
    extra = bytearray()
    ifd_pos = pos
    n = len(ents)
    ifd_len = (8 + n * 20 + 8) if bigtiff else (2 + n * 12 + 4)
    extra_pos = ifd_pos + ifd_len
    ifd = bytearray()
    ifd += struct.pack(E + ("Q" if bigtiff else "H"), n)
    for tag, typ, vals in ents:
        payload = struct.pack(E + fmt[typ] * len(vals), *vals)
        ifd += struct.pack(E + "HH" + ("Q" if bigtiff else "I"), tag, typ, len(vals))
        if len(payload) <= off_sz:
            ifd += payload + b"\0" * (off_sz - len(payload))
        else:
            ifd += struct.pack(E + ("Q" if bigtiff else "I"), extra_pos + len(extra))
            extra += payload
            if len(extra) & 1:
                extra += b"\0"
    ifd += struct.pack(E + ("Q" if bigtiff else "I"), 0)
    with open(path, "wb") as f:
        if bigtiff:
            f.write((b"MM" if big_endian else b"II") + struct.pack(E + "HHHQ", 43, 8, 0, ifd_pos))
        else:
            f.write((b"MM" if big_endian else b"II") + struct.pack(E + "HI", 42, ifd_pos))
        f.write(body)
        f.write(ifd)
        f.write(extra)


def write_block_shapefile(base, blocks, shape_type=15):
    """blocks: list of (id, minx, miny, maxx, maxy) -> base.shp/.shx/.dbf with PolygonZ
    rectangles and fields fid N(10), ID N(10), like blocks/esa_extent_blocks.*"""
    recs = []
    for (_id, x0, y0, x1, y1) in blocks:
        pts = [(x0, y1), (x1, y1), (x1, y0), (x0, y0), (x0, y1)]
        c = struct.pack("<i4d2i", shape_type, x0, y0, x1, y1, 1, 5) + struct.pack("<i", 0)
        c += b"".join(struct.pack("<2d", *p) for p in pts)
        if shape_type == 15:
            c += struct.pack("<2d", 0, 0) + struct.pack("<5d", *([0.0] * 5))
            c += struct.pack("<2d", 0, 0) + struct.pack("<5d", *([0.0] * 5))
        recs.append(c)
    total = 100 + sum(8 + len(c) for c in recs)
    xs = [b[1] for b in blocks] + [b[3] for b in blocks]
    ys = [b[2] for b in blocks] + [b[4] for b in blocks]

    def header(length_bytes):
        return (struct.pack(">i5ii", 9994, 0, 0, 0, 0, 0, length_bytes // 2) +
                struct.pack("<ii4d4d", 1000, shape_type, min(xs), min(ys), max(xs), max(ys), 0, 0, 0, 0))

    with open(base + ".shp", "wb") as f, open(base + ".shx", "wb") as fx:
        f.write(header(total))
        fx.write(header(100 + 8 * len(recs)))
        pos = 100
        for i, c in enumerate(recs):
            f.write(struct.pack(">ii", i + 1, len(c) // 2) + c)
            fx.write(struct.pack(">ii", pos // 2, len(c) // 2))
            pos += 8 + len(c)
    nrec = len(blocks)
    fields = [(b"fid", b"N", 10), (b"ID", b"N", 10)]
    hdr_len = 32 + 32 * len(fields) + 1
    rec_len = 1 + sum(f[2] for f in fields)
    with open(base + ".dbf", "wb") as f:
        f.write(struct.pack("<BBBBIHH20x", 3, 124, 1, 1, nrec, hdr_len, rec_len))
        for name, typ, ln in fields:
            f.write(name.ljust(11, b"\0") + typ + b"\0" * 4 + bytes([ln, 0]) + b"\0" * 14)
        f.write(b"\x0d")
        for i, b in enumerate(blocks):
            f.write(b" " + str(i + 1).rjust(10).encode() + str(b[0]).rjust(10).encode())
        f.write(b"\x1a")


def write_cog(path, img, gt=None, tile=(1024, 1024), overviews=2, compression=8, predictor=1, pixel_is_point=False):
    """A cloud-optimised-GeoTIFF-shaped file, as the ESA WorldCover tiles are
    (/root/reference/landcover/esa_worldcover_2021.vrt:265-272 names 36000^2 COGs with 1024^2 blocks):
    the full-resolution IFD first, `overviews` reduced-resolution IFDs (each half the size, NewSubfileType 1)
    chained behind it, ALL directories in front of the pixel data, and the tile data laid out overviews
    first (smallest level first), the full-resolution tiles last -- so a reader that takes "the first
    offsets it finds" or assumes directory order = data order reads an overview.  Classic little-endian TIFF."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    levels = [img]
    for _ in range(overviews):
        levels.append(np.ascontiguousarray(levels[-1][::2, ::2]))      # nearest-neighbour overview
    tw, th = tile

    def enc(c):
        if predictor == 2 and compression != 1:
            c = c.astype(np.int16)
            c[:, 1:] = c[:, 1:] - c[:, :-1]
            c = (c & 0xFF).astype(np.uint8)
        raw = c.tobytes()
        return raw if compression == 1 else zlib.compress(raw, 6)

    blobs = []
    for lv in levels:
        H, W = lv.shape
        bl = []
        for ty in range(0, H, th):
            for tx in range(0, W, tw):
                c = np.zeros((th, tw), np.uint8)
                part = lv[ty:ty + th, tx:tx + tw]
                c[:part.shape[0], :part.shape[1]] = part
                bl.append(enc(c))
        blobs.append(bl)

    SHORT, LONG, DOUBLE = 3, 4, 12
    fmt = {SHORT: "H", LONG: "I", DOUBLE: "d"}

    def ifd_bytes(level, ifd_pos, next_ifd, offsets):
        H, W = levels[level].shape
        ents = [(256, LONG, [W]), (257, LONG, [H]), (258, SHORT, [8]), (259, SHORT, [compression]),
                (262, SHORT, [1]), (277, SHORT, [1]), (284, SHORT, [1]),
                (322, SHORT, [tw]), (323, SHORT, [th]), (324, LONG, offsets), (325, LONG, [len(b) for b in blobs[level]])]
        if level > 0:
            ents.append((254, LONG, [1]))                  # NewSubfileType: reduced-resolution image
        if predictor != 1 and compression != 1:
            ents.append((317, SHORT, [predictor]))
        if gt is not None:
            s = 2 ** level
            ents.append((33550, DOUBLE, [gt[1] * s, -gt[5] * s, 0.0]))
            ents.append((33922, DOUBLE, [0.0, 0.0, 0.0, gt[0], gt[3], 0.0]))
        ents.append((34735, SHORT, [1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 2 if pixel_is_point else 1, 2048, 0, 1, 4326]))
        ents.sort(key=lambda e: e[0])
        n = len(ents)
        extra_pos = ifd_pos + 2 + n * 12 + 4
        ifd, extra = bytearray(struct.pack("<H", n)), bytearray()
        for tag, typ, vals in ents:
            payload = struct.pack("<" + fmt[typ] * len(vals), *vals)
            ifd += struct.pack("<HHI", tag, typ, len(vals))
            if len(payload) <= 4:
                ifd += payload + b"\0" * (4 - len(payload))
            else:
                ifd += struct.pack("<I", extra_pos + len(extra))
                extra += payload
                if len(extra) & 1:
                    extra += b"\0"
        ifd += struct.pack("<I", next_ifd)
        return bytes(ifd + extra)

    # pass 1: sizes of the directories; pass 2: the real offsets
    sizes = [len(ifd_bytes(l, 0, 0, [0] * len(blobs[l]))) for l in range(len(levels))]
    ifd_pos, pos = [], 8
    for sz in sizes:
        ifd_pos.append(pos)
        pos += sz + (sz & 1)
    offsets = [None] * len(levels)
    body = bytearray()
    for l in reversed(range(len(levels))):                  # overviews first, full resolution last
        offs = []
        for b in blobs[l]:
            offs.append(pos + len(body))
            body += b
            if len(body) & 1:
                body += b"\0"
        offsets[l] = offs
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_pos[0]))
        for l in range(len(levels)):
            d = ifd_bytes(l, ifd_pos[l], ifd_pos[l + 1] if l + 1 < len(levels) else 0, offsets[l])
            f.write(d + (b"\0" if len(d) & 1 else b""))
        f.write(body)
    return levels

/* tiff.c -- GeoTIFF reader and tiled-DEFLATE GeoTIFF writer without GDAL/libtiff.
 *
 * Replaces what the reference gets from GDAL in load_raster() / save_raster()
 * (/root/reference/src/raster.c:106-227) for the file kinds this program meets:
 *   read : 8-bit single-band (or pixel-interleaved, band 1 taken) TIFF, classic
 *          or BigTIFF, either byte order, strips or tiles, compression none /
 *          LZW (5) / Adobe DEFLATE (8, 32946) / PackBits (32773), predictor 1|2,
 *          georeferenced by ModelPixelScale + ModelTiepoint or ModelTransformation;
 *   write: GTiff, Byte, 1 band, COMPRESS=DEFLATE, TILED=YES (256 x 256 tiles),
 *          the creation options of src/raster.c:204-209, with the input's
 *          GeoKey tags copied so the projection is the landcover's
 *          (src/raster.c:212-214).  No NoData tag (the reference sets none).
 * Format sources: TIFF 6.0 (1992), BigTIFF design note, GeoTIFF 1.0 / OGC 19-008r4.
 * File *bytes* are not a parity target (GDAL's encoder is third party); decoded
 * pixels and geotransform are.
 */
#include "gcn10_host.h"
#include "host_internal.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

/* ------------------------------------------------------------------------ */
/* reader                                                                    */
/* ------------------------------------------------------------------------ */

enum { T_BYTE = 1, T_ASCII = 2, T_SHORT = 3, T_LONG = 4, T_RATIONAL = 5, T_SBYTE = 6,
       T_UNDEF = 7, T_SSHORT = 8, T_SLONG = 9, T_SRATIONAL = 10, T_FLOAT = 11,
       T_DOUBLE = 12, T_IFD = 13, T_LONG8 = 16, T_SLONG8 = 17, T_IFD8 = 18 };

struct gcn10_tiff {
    int fd;
    bool big, swap;
    uint32_t width, height;
    uint16_t bps, spp, compression, predictor, planar;
    bool tiled;
    uint32_t cw, ch;            /* chunk (tile or strip) width / height */
    uint32_t across, down;      /* chunks per row / column */
    uint64_t *offsets, *counts;
    uint64_t n_chunks;
    uint64_t file_size;
    uint64_t id_dev, id_ino;    /* which file (the chunk cache is shared by every open handle of it) */
    double gt[6];
    gcn10_georef georef;
};

static size_t type_size(unsigned t)
{
    switch (t) {
    case T_BYTE: case T_ASCII: case T_SBYTE: case T_UNDEF: return 1;
    case T_SHORT: case T_SSHORT: return 2;
    case T_LONG: case T_SLONG: case T_FLOAT: case T_IFD: return 4;
    case T_RATIONAL: case T_SRATIONAL: case T_DOUBLE: case T_LONG8: case T_SLONG8: case T_IFD8: return 8;
    default: return 0;
    }
}

static uint64_t rd(const unsigned char *p, int n, bool swap)
{
    uint64_t v = 0;

    if (swap)           /* file is big endian */
        for (int i = 0; i < n; i++)
            v = (v << 8) | p[i];
    else
        for (int i = n - 1; i >= 0; i--)
            v = (v << 8) | p[i];
    return v;
}

static int pread_all(int fd, void *buf, size_t n, uint64_t off)
{
    unsigned char *p = buf;

    while (n) {
        ssize_t r = pread(fd, p, n, (off_t)off);

        if (r < 0 && errno == EINTR)
            continue;
        if (r <= 0)
            return -1;
        p += r;
        off += (uint64_t)r;
        n -= (size_t)r;
    }
    return 0;
}

/* values of one directory entry as raw bytes (malloc'd), file byte order */
static unsigned char *entry_bytes(struct gcn10_tiff *t, const unsigned char *e, unsigned type,
                                  uint64_t count, size_t *nbytes)
{
    size_t ts = type_size(type);
    size_t inl = t->big ? 8 : 4;
    const unsigned char *valp = e + (t->big ? 12 : 8);
    unsigned char *buf;

    /* a value array can never be larger than the file that holds it */
    if (!ts || count > (1ull << 31) / ts || ts * count > t->file_size)
        return NULL;
    *nbytes = (size_t)(ts * count);
    buf = malloc(*nbytes ? *nbytes : 1);
    if (!buf)
        return NULL;
    if (*nbytes <= inl) {
        memcpy(buf, valp, *nbytes);
    }
    else {
        uint64_t off = rd(valp, (int)inl, t->swap);

        if (pread_all(t->fd, buf, *nbytes, off) != 0) {
            free(buf);
            return NULL;
        }
    }
    return buf;
}

static uint64_t *entry_u64(struct gcn10_tiff *t, const unsigned char *e, unsigned type, uint64_t count)
{
    size_t nb = 0, ts = type_size(type);
    unsigned char *raw = entry_bytes(t, e, type, count, &nb);
    uint64_t *out;

    if (!raw)
        return NULL;
    out = malloc((size_t)(count ? count : 1) * sizeof *out);
    if (out)
        for (uint64_t i = 0; i < count; i++)
            out[i] = rd(raw + i * ts, (int)ts, t->swap);
    free(raw);
    return out;
}

static double *entry_f64(struct gcn10_tiff *t, const unsigned char *e, unsigned type, uint64_t count)
{
    size_t nb = 0;
    unsigned char *raw;
    double *out;

    if (type != T_DOUBLE)
        return NULL;
    raw = entry_bytes(t, e, type, count, &nb);
    if (!raw)
        return NULL;
    out = malloc((size_t)(count ? count : 1) * sizeof *out);
    if (out)
        for (uint64_t i = 0; i < count; i++) {
            uint64_t v = rd(raw + i * 8, 8, t->swap);

            memcpy(&out[i], &v, 8);
        }
    free(raw);
    return out;
}

void gcn10_tiff_close_reader(struct gcn10_tiff *t)
{
    if (!t)
        return;
    if (t->fd >= 0)
        close(t->fd);
    free(t->offsets);
    free(t->counts);
    free(t->georef.geokeys);
    free(t->georef.geodoubles);
    free(t->georef.geoascii);
    free(t);
}

struct gcn10_tiff *gcn10_tiff_open_reader(const char *path, char *err, size_t errcap)
{
    struct gcn10_tiff *t = calloc(1, sizeof *t);
    unsigned char hdr[16], *dir = NULL;
    uint64_t ifd, nent;
    size_t esz;
    uint32_t rows_per_strip = 0, tw = 0, th = 0;
    double *scale = NULL, *tie = NULL, *xform = NULL;
    uint64_t n_scale = 0, n_tie = 0, n_xform = 0;
    bool have_off = false, have_cnt = false;

    if (!t) {
        snprintf(err, errcap, "out of memory for raster %s", path);
        return NULL;
    }
    t->fd = open(path, O_RDONLY);
    if (t->fd < 0 || pread_all(t->fd, hdr, 8, 0) != 0) {
        snprintf(err, errcap, "gdal open failed: %s", path);        /* src/raster.c:121 */
        goto fail;
    }
    {
        struct stat st;

        if (fstat(t->fd, &st) != 0 || st.st_size < 8)
            goto badfile;
        t->file_size = (uint64_t)st.st_size;
        t->id_dev = (uint64_t)st.st_dev;
        t->id_ino = (uint64_t)st.st_ino;
    }
    if (hdr[0] == 'I' && hdr[1] == 'I')
        t->swap = false;
    else if (hdr[0] == 'M' && hdr[1] == 'M')
        t->swap = true;
    else {
        snprintf(err, errcap, "gdal open failed: %s (not a TIFF)", path);
        goto fail;
    }
    {
        unsigned magic = (unsigned)rd(hdr + 2, 2, t->swap);

        if (magic == 42) {
            ifd = rd(hdr + 4, 4, t->swap);
        }
        else if (magic == 43) {
            t->big = true;
            if (pread_all(t->fd, hdr, 16, 0) != 0)
                goto badfile;
            ifd = rd(hdr + 8, 8, t->swap);
        }
        else {
            goto badfile;
        }
    }
    {
        unsigned char cnt[8];
        int cn = t->big ? 8 : 2;

        if (pread_all(t->fd, cnt, (size_t)cn, ifd) != 0)
            goto badfile;
        nent = rd(cnt, cn, t->swap);
        esz = t->big ? 20 : 12;
        if (nent == 0 || nent > 4096)
            goto badfile;
        dir = malloc((size_t)nent * esz);
        if (!dir || pread_all(t->fd, dir, (size_t)nent * esz, ifd + (uint64_t)cn) != 0)
            goto badfile;
    }
    t->bps = 1;
    t->spp = 1;
    t->compression = 1;
    t->predictor = 1;
    t->planar = 1;
    for (uint64_t i = 0; i < nent; i++) {
        const unsigned char *e = dir + i * esz;
        unsigned tag = (unsigned)rd(e, 2, t->swap);
        unsigned type = (unsigned)rd(e + 2, 2, t->swap);
        uint64_t count = t->big ? rd(e + 4, 8, t->swap) : rd(e + 4, 4, t->swap);
        uint64_t *v = NULL;

        switch (tag) {
        case 256: case 257: case 258: case 259: case 277: case 278: case 284: case 317:
        case 322: case 323:
            v = entry_u64(t, e, type, count);
            if (!v || count < 1) {
                free(v);
                goto badfile;
            }
            if (tag == 256) t->width = (uint32_t)v[0];
            else if (tag == 257) t->height = (uint32_t)v[0];
            else if (tag == 258) t->bps = (uint16_t)v[0];
            else if (tag == 259) t->compression = (uint16_t)v[0];
            else if (tag == 277) t->spp = (uint16_t)v[0];
            else if (tag == 278) rows_per_strip = (uint32_t)v[0];
            else if (tag == 284) t->planar = (uint16_t)v[0];
            else if (tag == 317) t->predictor = (uint16_t)v[0];
            else if (tag == 322) tw = (uint32_t)v[0];
            else if (tag == 323) th = (uint32_t)v[0];
            free(v);
            break;
        case 273: case 324:         /* StripOffsets / TileOffsets */
            free(t->offsets);
            t->offsets = entry_u64(t, e, type, count);
            t->n_chunks = count;
            have_off = t->offsets != NULL;
            if (tag == 324)
                t->tiled = true;
            break;
        case 279: case 325:         /* StripByteCounts / TileByteCounts */
            free(t->counts);
            t->counts = entry_u64(t, e, type, count);
            have_cnt = t->counts != NULL;
            break;
        case 33550:
            scale = entry_f64(t, e, type, count);
            n_scale = count;
            break;
        case 33922:
            tie = entry_f64(t, e, type, count);
            n_tie = count;
            break;
        case 34264:
            xform = entry_f64(t, e, type, count);
            n_xform = count;
            break;
        case 34735:
            v = entry_u64(t, e, type, count);
            if (v) {
                t->georef.geokeys = malloc((size_t)(count ? count : 1) * sizeof(uint16_t));
                if (t->georef.geokeys) {
                    for (uint64_t k = 0; k < count; k++)
                        t->georef.geokeys[k] = (uint16_t)v[k];
                    t->georef.n_geokeys = (int)count;
                }
                free(v);
            }
            break;
        case 34736:
            t->georef.geodoubles = entry_f64(t, e, type, count);
            t->georef.n_geodoubles = t->georef.geodoubles ? (int)count : 0;
            break;
        case 34737: {
            size_t nb = 0;
            unsigned char *raw = entry_bytes(t, e, T_ASCII, count, &nb);

            if (raw) {
                t->georef.geoascii = malloc(nb + 1);
                if (t->georef.geoascii) {
                    memcpy(t->georef.geoascii, raw, nb);
                    t->georef.geoascii[nb] = '\0';
                }
                free(raw);
            }
            break;
        }
        default:
            break;
        }
    }
    free(dir);
    dir = NULL;

    if (!t->width || !t->height || !have_off || !have_cnt)
        goto badfile;
    if (t->bps != 8) {
        snprintf(err, errcap, "gdal open failed: %s (%u bits per sample; only Byte rasters are supported)",
                 path, t->bps);
        goto fail;
    }
    if (t->planar != 1 && t->spp != 1) {
        /* band-sequential: band 1 is the first n_chunks / spp chunks */
        t->n_chunks /= t->spp;
        t->spp = 1;
    }
    if (t->tiled) {
        if (!tw || !th)
            goto badfile;
        t->cw = tw;
        t->ch = th;
    }
    else {
        t->cw = t->width;
        t->ch = rows_per_strip && rows_per_strip < t->height ? rows_per_strip : t->height;
    }
    t->across = (t->width + t->cw - 1) / t->cw;
    t->down = (t->height + t->ch - 1) / t->ch;
    if ((uint64_t)t->across * t->down > t->n_chunks)
        goto badfile;
    if (t->spp == 0 || t->spp > 16 || (uint64_t)t->cw * t->ch * t->spp > (1ull << 30) ||
        t->width > 0x7fffffffu || t->height > 0x7fffffffu)
        goto badfile;
    switch (t->compression) {
    case 1: case 5: case 8: case 32946: case 32773:
        break;
    default:
        snprintf(err, errcap, "gdal open failed: %s (TIFF compression %u not supported)", path,
                 t->compression);
        goto fail;
    }

    /* geotransform in GDAL order {x0, dx, rx, y0, ry, dy} */
    t->gt[0] = 0; t->gt[1] = 1; t->gt[2] = 0; t->gt[3] = 0; t->gt[4] = 0; t->gt[5] = 1;
    if (xform && n_xform >= 16) {
        t->gt[0] = xform[3]; t->gt[1] = xform[0]; t->gt[2] = xform[1];
        t->gt[3] = xform[7]; t->gt[4] = xform[4]; t->gt[5] = xform[5];
    }
    else if (scale && n_scale >= 2 && tie && n_tie >= 6) {
        /* GDAL: origin = tiepoint world - tiepoint pixel * scale; north-up */
        t->gt[1] = scale[0];
        t->gt[5] = -scale[1];
        t->gt[0] = tie[3] - tie[0] * scale[0];
        t->gt[3] = tie[4] + tie[1] * scale[1];
    }
    /* RasterPixelIsPoint (GTRasterTypeGeoKey 1025 = 2) shifts by half a pixel, as GDAL does */
    if (t->georef.geokeys && t->georef.n_geokeys >= 4) {
        int nk = t->georef.geokeys[3];

        for (int k = 0; k < nk && 4 + 4 * k + 3 < t->georef.n_geokeys; k++) {
            const uint16_t *key = t->georef.geokeys + 4 + 4 * k;

            if (key[0] == 1025 && key[1] == 0 && key[3] == 2) {
                t->gt[0] -= 0.5 * t->gt[1] + 0.5 * t->gt[2];
                t->gt[3] -= 0.5 * t->gt[4] + 0.5 * t->gt[5];
            }
        }
    }
    free(scale);
    free(tie);
    free(xform);
    return t;

badfile:
    snprintf(err, errcap, "gdal open failed: %s (unreadable or unsupported TIFF structure)", path);
fail:
    free(dir);
    free(scale);
    free(tie);
    free(xform);
    gcn10_tiff_close_reader(t);
    return NULL;
}

void gcn10_tiff_reader_info(const struct gcn10_tiff *t, int *xsize, int *ysize, double gt[6])
{
    *xsize = (int)t->width;
    *ysize = (int)t->height;
    memcpy(gt, t->gt, sizeof t->gt);
}

const gcn10_georef *gcn10_tiff_reader_georef(const struct gcn10_tiff *t)
{
    return &t->georef;
}

/* TIFF LZW (TIFF 6.0 section 13): MSB-first codes of 9..12 bits, ClearCode 256,
 * EndOfInformation 257, "early change" of the code width. */
static int lzw_decode(const unsigned char *src, size_t n, unsigned char *dst, size_t cap)
{
    enum { CLEAR = 256, EOI = 257, FIRST = 258, MAXC = 4096 };
    static __thread uint16_t prefix[MAXC];
    static __thread unsigned char suffix[MAXC], first[MAXC];
    static __thread uint16_t length[MAXC];
    uint32_t bits = 0;
    int nbits = 0, width = 9, next = FIRST, prev = -1;
    size_t ip = 0, op = 0;

    for (int i = 0; i < 256; i++) {
        prefix[i] = 0;
        suffix[i] = first[i] = (unsigned char)i;
        length[i] = 1;
    }
    for (;;) {
        int code;

        while (nbits < width) {
            if (ip >= n)
                return op == cap ? 0 : -1;      /* stream ended without EOI */
            bits = (bits << 8) | src[ip++];
            nbits += 8;
        }
        code = (int)((bits >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == EOI)
            break;
        if (code == CLEAR) {
            width = 9;
            next = FIRST;
            prev = -1;
            continue;
        }
        if (prev < 0) {
            if (code >= 256 || op >= cap)
                return op >= cap ? 0 : -1;
            dst[op++] = (unsigned char)code;
            prev = code;
            continue;
        }
        {
            /* the string of `code`; code == next is the KwKwK case: string(prev) + its
             * own first byte */
            const bool kwkwk = code == next;
            size_t full, pos;
            unsigned char fc;
            int c;

            if (code > next || (kwkwk && next >= MAXC))
                return -1;
            fc = kwkwk ? first[prev] : first[code];
            full = kwkwk ? (size_t)length[prev] + 1 : (size_t)length[code];
            pos = full;
            c = kwkwk ? prev : code;
            if (kwkwk) {
                pos--;
                if (op + pos < cap)
                    dst[op + pos] = fc;
            }
            while (pos > 0) {           /* strings are stored as (prefix, last byte) */
                pos--;
                if (op + pos < cap)
                    dst[op + pos] = suffix[c];
                c = prefix[c];
            }
            op += full;
            if (next < MAXC) {
                prefix[next] = (uint16_t)prev;
                suffix[next] = fc;
                first[next] = first[prev];
                length[next] = (uint16_t)(length[prev] + 1);
                next++;
                if (next + 1 >= (1 << width) && width < 12)    /* early change */
                    width++;
            }
            prev = code;
            if (op >= cap)
                return 0;               /* chunk complete; trailing codes are padding */
        }
    }
    return 0;
}

static int packbits_decode(const unsigned char *src, size_t n, unsigned char *dst, size_t cap)
{
    size_t ip = 0, op = 0;

    while (ip < n && op < cap) {
        int c = (signed char)src[ip++];

        if (c >= 0) {
            size_t len = (size_t)c + 1;

            if (ip + len > n)
                return -1;
            if (op + len > cap)
                len = cap - op;
            memcpy(dst + op, src + ip, len);
            ip += (size_t)c + 1;
            op += len;
        }
        else if (c != -128) {
            size_t len = (size_t)(1 - c);

            if (ip >= n)
                return -1;
            if (op + len > cap)
                len = cap - op;
            memset(dst + op, src[ip++], len);
            op += len;
        }
    }
    return 0;
}

/* decodes the first `need` bytes of chunk `idx` into buf (the chunk holds cw*rows*spp bytes;
 * strips may be shorter at the end).  A window that ends early in a chunk -- a few columns of
 * a full-width strip, as in the soil raster -- does not pay for the rest of it. */
static int decode_chunk(struct gcn10_tiff *t, uint64_t idx, unsigned char *buf, size_t rawcap,
                        unsigned char **scratch, size_t *scratch_cap, uint32_t rows_in_chunk, size_t need)
{
    uint64_t off = t->offsets[idx], cnt = t->counts[idx];
    size_t want = (size_t)t->cw * rows_in_chunk * t->spp;

    if (want > rawcap)
        return -1;
    if (need < want)
        want = need;
    if (off > t->file_size || cnt > t->file_size - off)
        return -1;                  /* chunk outside the file: corrupt directory */
    if (cnt == 0) {                 /* sparse tile: GDAL reads it as zeros */
        memset(buf, 0, want);
        return 0;
    }
    if (t->compression == 1) {
        if (cnt < want)
            return -1;
        return pread_all(t->fd, buf, want, off);
    }
    if (cnt > *scratch_cap) {
        unsigned char *g = realloc(*scratch, (size_t)cnt);

        if (!g)
            return -1;
        *scratch = g;
        *scratch_cap = (size_t)cnt;
    }
    if (pread_all(t->fd, *scratch, (size_t)cnt, off) != 0)
        return -1;
    if (t->compression == 8 || t->compression == 32946) {
        uLongf dl = (uLongf)want;
        int rc = uncompress(buf, &dl, *scratch, (uLong)cnt);

        /* Z_BUF_ERROR with a full buffer = stream longer than the chunk (padding) */
        if (rc != Z_OK && !(rc == Z_BUF_ERROR && dl == want))
            return -1;
        if (dl < want)
            memset(buf + dl, 0, want - dl);
    }
    else if (t->compression == 5) {
        if (lzw_decode(*scratch, (size_t)cnt, buf, want) != 0)
            return -1;
    }
    else {
        if (packbits_decode(*scratch, (size_t)cnt, buf, want) != 0)
            return -1;
    }
    if (t->predictor == 2) {        /* horizontal differencing, per row, per sample */
        size_t rowb = (size_t)t->cw * t->spp;

        for (uint32_t r = 0; r < rows_in_chunk && (size_t)r * rowb < want; r++) {
            unsigned char *row = buf + (size_t)r * rowb;
            size_t end = want - (size_t)r * rowb < rowb ? want - (size_t)r * rowb : rowb;

            for (size_t i = t->spp; i < end; i++)
                row[i] = (unsigned char)(row[i] + row[i - t->spp]);
        }
    }
    return 0;
}

/* Decoded chunks that windows use a small part of.  The soil raster is a global file of full-width strips: a
 * block's window needs 1/120 of each strip it touches, and the ~120 blocks of a latitude band need the SAME strips,
 * one block after the other (the queue hands out ids in shapefile order).  Decoding a strip once per block was the
 * largest item of the host's CPU time without the file sink (profiles/r03/host_cpu_by_thread_and_job.txt).
 * One cache per process, shared by every handle of a file (workers open their own), least recently used out first;
 * GCN10_CHUNK_CACHE_MB (default 512, 0 = none).  Entries in use (being copied from) are not evicted. */
struct cache_entry {
    uint64_t dev, ino, idx;
    unsigned char *data;
    size_t bytes;
    uint64_t stamp;
    int refs;
};

static struct {
    pthread_mutex_t mu;
    struct cache_entry **e;     /* heap objects: a pinned entry is referred to by pointer while others come and go */
    int n, cap_entries;
    size_t bytes, cap_bytes;
    uint64_t clock, hits, misses;
    bool ready;
} g_cache = { .mu = PTHREAD_MUTEX_INITIALIZER };

static void cache_init_locked(void)
{
    const char *mb = getenv("GCN10_CHUNK_CACHE_MB");

    g_cache.cap_bytes = (size_t)(mb ? strtoull(mb, NULL, 10) : 512) << 20;
    g_cache.ready = true;
}

/* the chunk's decoded bytes (pinned until cache_release), or NULL */
static struct cache_entry *cache_get(const struct gcn10_tiff *t, uint64_t idx, size_t bytes)
{
    struct cache_entry *hit = NULL;

    pthread_mutex_lock(&g_cache.mu);
    if (!g_cache.ready)
        cache_init_locked();
    for (int i = 0; i < g_cache.n; i++) {
        struct cache_entry *c = g_cache.e[i];

        if (c->idx == idx && c->ino == t->id_ino && c->dev == t->id_dev && c->bytes == bytes) {
            c->refs++;
            c->stamp = ++g_cache.clock;
            hit = c;
            break;
        }
    }
    if (hit)
        g_cache.hits++;
    else
        g_cache.misses++;
    pthread_mutex_unlock(&g_cache.mu);
    return hit;
}

static void cache_release(struct cache_entry *c)
{
    pthread_mutex_lock(&g_cache.mu);
    c->refs--;
    pthread_mutex_unlock(&g_cache.mu);
}

/* hands `data` (malloc'ed, `bytes` long) to the cache; returns the entry to read from (pinned), or NULL when the cache
 * does not take it (the caller keeps and frees `data`) */
static struct cache_entry *cache_put(const struct gcn10_tiff *t, uint64_t idx, unsigned char *data, size_t bytes)
{
    struct cache_entry *res = NULL;

    pthread_mutex_lock(&g_cache.mu);
    if (bytes <= g_cache.cap_bytes / 4) {
        for (int i = 0; i < g_cache.n; i++) {       /* decoded by another thread meanwhile: use theirs */
            struct cache_entry *c = g_cache.e[i];

            if (c->idx == idx && c->ino == t->id_ino && c->dev == t->id_dev && c->bytes == bytes) {
                c->refs++;
                pthread_mutex_unlock(&g_cache.mu);
                free(data);
                return c;
            }
        }
        while (g_cache.bytes + bytes > g_cache.cap_bytes) {      /* least recently used, not in use, out */
            int victim = -1;

            for (int i = 0; i < g_cache.n; i++)
                if (g_cache.e[i]->refs == 0 && (victim < 0 || g_cache.e[i]->stamp < g_cache.e[victim]->stamp))
                    victim = i;
            if (victim < 0)
                break;
            g_cache.bytes -= g_cache.e[victim]->bytes;
            free(g_cache.e[victim]->data);
            free(g_cache.e[victim]);
            g_cache.e[victim] = g_cache.e[--g_cache.n];
        }
        if (g_cache.bytes + bytes <= g_cache.cap_bytes) {
            if (g_cache.n == g_cache.cap_entries) {
                int cap = g_cache.cap_entries ? g_cache.cap_entries * 2 : 4096;
                struct cache_entry **g = realloc(g_cache.e, (size_t)cap * sizeof *g);

                if (g) {
                    g_cache.e = g;
                    g_cache.cap_entries = cap;
                }
            }
            if (g_cache.n < g_cache.cap_entries && (res = malloc(sizeof *res)) != NULL) {
                *res = (struct cache_entry){ t->id_dev, t->id_ino, idx, data, bytes, ++g_cache.clock, 1 };
                g_cache.e[g_cache.n++] = res;
                g_cache.bytes += bytes;
            }
        }
    }
    pthread_mutex_unlock(&g_cache.mu);
    return res;
}

static bool cache_enabled(void)
{
    bool on;

    pthread_mutex_lock(&g_cache.mu);
    if (!g_cache.ready)
        cache_init_locked();
    on = g_cache.cap_bytes > 0;
    pthread_mutex_unlock(&g_cache.mu);
    return on;
}

void gcn10_tiff_cache_stats(uint64_t *hits, uint64_t *misses, size_t *bytes)
{
    pthread_mutex_lock(&g_cache.mu);
    if (hits)
        *hits = g_cache.hits;
    if (misses)
        *misses = g_cache.misses;
    if (bytes)
        *bytes = g_cache.bytes;
    pthread_mutex_unlock(&g_cache.mu);
}

/* one chunk (tile or strip) of a window read: decode and copy the overlap */
struct chunk_job {
    struct gcn10_tiff *t;
    uint32_t cx, cy;
    int xoff, yoff, xcount, ycount;
    uint8_t *dst;
    size_t dst_stride;
    /* completion */
    pthread_mutex_t *mu;
    pthread_cond_t *cv;
    int *pending;
    int *failed;
};

static int read_chunk(const struct chunk_job *j)
{
    /* per-thread decode buffers, reused across chunks */
    static __thread unsigned char *raw = NULL, *scratch = NULL;
    static __thread size_t raw_cap = 0, scratch_cap = 0;
    struct gcn10_tiff *t = j->t;
    size_t rawcap = (size_t)t->cw * t->ch * t->spp;
    uint32_t y_lo = j->cy * t->ch, x_lo = j->cx * t->cw;
    uint32_t rows_in_chunk = t->tiled ? t->ch : (y_lo + t->ch <= t->height ? t->ch : t->height - y_lo);
    uint32_t ys = (uint32_t)j->yoff > y_lo ? (uint32_t)j->yoff : y_lo;
    uint32_t ye = (uint32_t)(j->yoff + j->ycount) < y_lo + rows_in_chunk ? (uint32_t)(j->yoff + j->ycount)
                                                                         : y_lo + rows_in_chunk;
    uint32_t xs = (uint32_t)j->xoff > x_lo ? (uint32_t)j->xoff : x_lo;
    uint32_t xe = (uint32_t)(j->xoff + j->xcount) < x_lo + t->cw ? (uint32_t)(j->xoff + j->xcount)
                                                                 : x_lo + t->cw;

    if (rawcap > raw_cap) {
        unsigned char *g = realloc(raw, rawcap);

        if (!g)
            return -1;
        raw = g;
        raw_cap = rawcap;
    }
    if (xs >= xe || ys >= ye)
        return 0;
    {
        const uint64_t idx = (uint64_t)j->cy * t->across + j->cx;
        const size_t chunk_bytes = (size_t)t->cw * rows_in_chunk * t->spp;
        const unsigned char *from = raw;
        struct cache_entry *held = NULL;

        /* an uncompressed chunk the window uses at most a quarter of (full-width strips): the wanted part of every
         * row straight from the file -- reading the chunk from its start to the last wanted pixel copied ~120 times
         * the bytes a block of a global raster needs */
        if (t->compression == 1 && t->predictor != 2 && t->spp == 1 &&
            (size_t)(xe - xs) * (ye - ys) * 4 <= chunk_bytes) {
            const uint64_t off = t->offsets[idx], cnt = t->counts[idx];

            if (cnt == 0) {
                for (uint32_t y = ys; y < ye; y++)
                    memset(j->dst + (size_t)(y - (uint32_t)j->yoff) * j->dst_stride + (xs - (uint32_t)j->xoff), 0, xe - xs);
                return 0;
            }
            if (off > t->file_size || cnt > t->file_size - off ||
                cnt < ((size_t)(ye - 1 - y_lo) * t->cw + (xe - x_lo)))
                return -1;
            for (uint32_t y = ys; y < ye; y++)
                if (pread_all(t->fd, j->dst + (size_t)(y - (uint32_t)j->yoff) * j->dst_stride + (xs - (uint32_t)j->xoff),
                              xe - xs, off + (uint64_t)(y - y_lo) * t->cw + (xs - x_lo)) != 0)
                    return -1;
            return 0;
        }
        /* a compressed chunk the window uses at most a quarter of: through the cache, decoded in full */
        if (t->compression != 1 && (size_t)(xe - xs) * (ye - ys) * t->spp * 4 <= chunk_bytes &&
            cache_enabled()) {
            held = cache_get(t, idx, chunk_bytes);
            if (!held) {
                unsigned char *whole = malloc(chunk_bytes);

                if (whole && decode_chunk(t, idx, whole, chunk_bytes, &scratch, &scratch_cap, rows_in_chunk, chunk_bytes) == 0) {
                    held = cache_put(t, idx, whole, chunk_bytes);
                    if (!held) {            /* not taken: copy from our own buffer below, then free it */
                        memcpy(raw, whole, chunk_bytes <= rawcap ? chunk_bytes : rawcap);
                        free(whole);
                    }
                }
                else {
                    free(whole);
                    return -1;
                }
            }
            if (held)
                from = held->data;
        }
        else if (decode_chunk(t, idx, raw, rawcap, &scratch, &scratch_cap, rows_in_chunk,
                              ((size_t)(ye - 1 - y_lo) * t->cw + (xe - x_lo)) * t->spp) != 0) {
            return -1;
        }
        for (uint32_t y = ys; y < ye; y++) {
            const unsigned char *s = from + ((size_t)(y - y_lo) * t->cw + (xs - x_lo)) * t->spp;
            uint8_t *d = j->dst + (size_t)(y - (uint32_t)j->yoff) * j->dst_stride + (xs - (uint32_t)j->xoff);

            if (t->spp == 1) {
                memcpy(d, s, xe - xs);
            }
            else {
                for (uint32_t x = 0; x < xe - xs; x++)
                    d[x] = s[(size_t)x * t->spp];
            }
        }
        if (held)
            cache_release(held);
    }
    return 0;
}

static void chunk_job_run(void *arg)
{
    struct chunk_job *j = arg;
    int rc = read_chunk(j);

    pthread_mutex_lock(j->mu);
    if (rc != 0)
        *j->failed = 1;
    if (--*j->pending == 0)
        pthread_cond_broadcast(j->cv);
    pthread_mutex_unlock(j->mu);
    free(j);
}

int gcn10_tiff_read_window_mt(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount,
                              uint8_t *dst, size_t dst_stride, gcn10_pool *pool, char *err,
                              size_t errcap)
{
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    int pending = 0, failed = 0;

    if (xoff < 0 || yoff < 0 || xcount <= 0 || ycount <= 0 ||
        (uint64_t)xoff + (uint64_t)xcount > t->width || (uint64_t)yoff + (uint64_t)ycount > t->height) {
        snprintf(err, errcap, "window %d,%d %dx%d outside raster %ux%u", xoff, yoff, xcount, ycount,
                 t->width, t->height);
        return -1;
    }
    for (uint32_t cy = (uint32_t)yoff / t->ch; cy <= (uint32_t)(yoff + ycount - 1) / t->ch; cy++) {
        for (uint32_t cx = (uint32_t)xoff / t->cw; cx <= (uint32_t)(xoff + xcount - 1) / t->cw; cx++) {
            struct chunk_job job = { t, cx, cy, xoff, yoff, xcount, ycount, dst, dst_stride,
                                     &mu, &cv, &pending, &failed };
            struct chunk_job *j = pool ? malloc(sizeof *j) : NULL;

            if (!j) {               /* no pool (or no memory for the job): decode here */
                if (read_chunk(&job) != 0)
                    failed = 1;
                continue;
            }
            *j = job;
            pthread_mutex_lock(&mu);
            pending++;
            pthread_mutex_unlock(&mu);
            gcn10_pool_submit(pool, chunk_job_run, j);
        }
    }
    pthread_mutex_lock(&mu);
    while (pending > 0)
        pthread_cond_wait(&cv, &mu);
    pthread_mutex_unlock(&mu);
    if (failed) {
        snprintf(err, errcap, "gdalrasterio error: cannot decode a %s of the window %d,%d %dx%d",
                 t->tiled ? "tile" : "strip", xoff, yoff, xcount, ycount);
        return -1;
    }
    return 0;
}

int gcn10_tiff_read_window(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount,
                           uint8_t *dst, size_t dst_stride, char *err, size_t errcap)
{
    return gcn10_tiff_read_window_mt(t, xoff, yoff, xcount, ycount, dst, dst_stride, NULL, err, errcap);
}

/* The chunks of a window as they lie in the file, for the GPU side: DEFLATE chunks are inflated
 * there, uncompressed chunks are only untiled there, and TIFF predictor 2 (horizontal differencing)
 * is undone there in either case.  LZW and PackBits stay with the host reader (return 1).  Mirrors
 * the clipping of read_chunk().  Of an uncompressed chunk only the bytes from the first wanted pixel
 * to the last wanted pixel are staged. */
int gcn10_tiff_plan_window(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount, int dst_x,
                           int dst_y, struct gcn10_read_plan *plan, char *err, size_t errcap)
{
    const bool raw = t->compression == 1;
    const bool deflate = t->compression == 8 || t->compression == 32946;

    if ((!raw && !deflate) || (t->predictor != 1 && t->predictor != 2) || t->spp != 1 ||
        t->bps != 8 || (uint64_t)t->cw * t->ch > ((uint64_t)1 << 28))
        return 1;
    if (xoff < 0 || yoff < 0 || xcount <= 0 || ycount <= 0 ||
        (uint64_t)xoff + (uint64_t)xcount > t->width || (uint64_t)yoff + (uint64_t)ycount > t->height) {
        snprintf(err, errcap, "window %d,%d %dx%d outside raster %ux%u", xoff, yoff, xcount, ycount,
                 t->width, t->height);
        return -1;
    }
    for (uint32_t cy = (uint32_t)yoff / t->ch; cy <= (uint32_t)(yoff + ycount - 1) / t->ch; cy++) {
        for (uint32_t cx = (uint32_t)xoff / t->cw; cx <= (uint32_t)(xoff + xcount - 1) / t->cw; cx++) {
            uint64_t idx = (uint64_t)cy * t->across + cx;
            uint64_t off = t->offsets[idx], cnt = t->counts[idx];
            uint32_t y_lo = cy * t->ch, x_lo = cx * t->cw;
            uint32_t rows = t->tiled ? t->ch : (y_lo + t->ch <= t->height ? t->ch : t->height - y_lo);
            uint32_t ys = (uint32_t)yoff > y_lo ? (uint32_t)yoff : y_lo;
            uint32_t ye = (uint32_t)(yoff + ycount) < y_lo + rows ? (uint32_t)(yoff + ycount) : y_lo + rows;
            uint32_t xs = (uint32_t)xoff > x_lo ? (uint32_t)xoff : x_lo;
            uint32_t xe = (uint32_t)(xoff + xcount) < x_lo + t->cw ? (uint32_t)(xoff + xcount) : x_lo + t->cw;
            struct gcn10_chunk_ref *c;

            if (cnt == 0 || xs >= xe || ys >= ye)
                continue;               /* sparse chunk: reads as zeros */
            if (off > t->file_size || cnt > t->file_size - off) {
                snprintf(err, errcap, "gdalrasterio error: a %s of the window %d,%d %dx%d lies outside the file",
                         t->tiled ? "tile" : "strip", xoff, yoff, xcount, ycount);
                return -1;
            }
            if (cnt > 0x7fffffffu)
                return 1;
            if (plan->n == plan->cap) {
                size_t cap = plan->cap ? plan->cap * 2 : 256;
                struct gcn10_chunk_ref *g = realloc(plan->chunks, cap * sizeof *g);

                if (!g) {
                    snprintf(err, errcap, "out of memory for the read plan");
                    return -1;
                }
                plan->chunks = g;
                plan->cap = cap;
            }
            c = &plan->chunks[plan->n];
            c->fd = t->fd;
            c->file_off = off;
            c->nbytes = (uint32_t)cnt;
            c->chunk_w = t->cw;
            c->rows = rows;
            c->src_x = xs - x_lo;
            c->src_y = ys - y_lo;
            c->copy_w = xe - xs;
            c->copy_h = ye - ys;
            c->dst_x = (uint32_t)dst_x + (xs - (uint32_t)xoff);
            c->dst_y = (uint32_t)dst_y + (ys - (uint32_t)yoff);
            /* (the Predictor tag belongs to the LZW / DEFLATE codecs: libtiff, and with it GDAL, ignores it on
             * uncompressed data, and so does decode_chunk()) */
            c->flags = raw ? GCN10_TILE_RAW : (t->predictor == 2 ? GCN10_TILE_PREDICTOR2 : 0u);
            c->out_len = t->cw * rows;
            if (raw) {
                /* the bytes that matter: from the first wanted pixel to the last one */
                const uint64_t first = (uint64_t)c->src_y * t->cw + c->src_x;
                const uint64_t last = (uint64_t)(c->src_y + c->copy_h - 1u) * t->cw + c->src_x + c->copy_w;

                if (cnt < last) {       /* a chunk shorter than its pixels: what decode_chunk() refuses */
                    snprintf(err, errcap, "gdalrasterio error: cannot decode a %s of the window %d,%d %dx%d",
                             t->tiled ? "tile" : "strip", xoff, yoff, xcount, ycount);
                    return -1;
                }
                /* full-width strips of a raster much wider than the window: most staged bytes would be
                 * other blocks' pixels -- the host reader copies rows instead */
                if (last - first > 4u * (uint64_t)c->copy_w * c->copy_h + ((uint64_t)1 << 20))
                    return 1;
                c->file_off = off + first;
                c->nbytes = (uint32_t)(last - first);
                c->out_len = c->nbytes;
                c->src_y = 0;
                c->src_x = 0;
            }
            else if (t->cw * rows > plan->max_chunk_bytes) {
                plan->max_chunk_bytes = t->cw * rows;
            }
            plan->n++;
            plan->staged_bytes += c->nbytes;
            plan->covered += (uint64_t)c->copy_w * c->copy_h;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* writer                                                                    */
/* ------------------------------------------------------------------------ */

enum { TILE = 256 };        /* GDAL's default block size for TILED=YES */

struct gcn10_tiff_writer {
    int fd;
    char *path;
    char *part;                 /* the file is written under this name and renamed when complete */
    int xsize, ysize, across, down;
    double gt[6];
    gcn10_georef georef;        /* deep copy */
    uint32_t *offsets, *counts;
    uint64_t pos;               /* append position */
    pthread_mutex_t mu;
    bool failed;
    bool direct;                /* O_DIRECT is set on fd: extents go out at 4096-aligned positions */
};

static int write_all(int fd, const void *buf, size_t n, uint64_t off)
{
    const unsigned char *p = buf;

    while (n) {
        ssize_t w = pwrite(fd, p, n, (off_t)off);

        if (w < 0 && errno == EINTR)
            continue;
        if (w <= 0)
            return -1;
        p += w;
        off += (uint64_t)w;
        n -= (size_t)w;
    }
    return 0;
}

static void free_georef(gcn10_georef *g)
{
    free(g->geokeys);
    free(g->geodoubles);
    free(g->geoascii);
    memset(g, 0, sizeof *g);
}

static int copy_georef(gcn10_georef *dst, const gcn10_georef *src)
{
    /* default: EPSG:4326 geographic, pixel-is-area -- what both inputs of this
     * program are (README of the reference: "HYSOGs250m_4326", WorldCover) */
    static const uint16_t wgs84[] = { 1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326 };

    memset(dst, 0, sizeof *dst);
    if (!src || !src->geokeys || src->n_geokeys < 4) {
        dst->geokeys = malloc(sizeof wgs84);
        if (!dst->geokeys)
            return -1;
        memcpy(dst->geokeys, wgs84, sizeof wgs84);
        dst->n_geokeys = (int)(sizeof wgs84 / sizeof wgs84[0]);
        return 0;
    }
    dst->geokeys = malloc((size_t)src->n_geokeys * sizeof(uint16_t));
    if (!dst->geokeys)
        return -1;
    memcpy(dst->geokeys, src->geokeys, (size_t)src->n_geokeys * sizeof(uint16_t));
    dst->n_geokeys = src->n_geokeys;
    if (src->geodoubles && src->n_geodoubles > 0) {
        dst->geodoubles = malloc((size_t)src->n_geodoubles * sizeof(double));
        if (!dst->geodoubles)
            return -1;
        memcpy(dst->geodoubles, src->geodoubles, (size_t)src->n_geodoubles * sizeof(double));
        dst->n_geodoubles = src->n_geodoubles;
    }
    if (src->geoascii) {
        dst->geoascii = strdup(src->geoascii);
        if (!dst->geoascii)
            return -1;
    }
    return 0;
}

gcn10_tiff_writer *gcn10_tiff_create(const char *path, int xsize, int ysize, const double gt[6],
                                     const gcn10_georef *georef, char *err, size_t errcap)
{
    static const unsigned char header[8] = { 'I', 'I', 42, 0, 0, 0, 0, 0 };
    gcn10_tiff_writer *w;
    size_t nt;

    if (xsize <= 0 || ysize <= 0) {
        snprintf(err, errcap, "write error: bad raster size %dx%d for %s", xsize, ysize, path);
        return NULL;
    }
    w = calloc(1, sizeof *w);
    if (!w)
        goto oom;
    w->fd = -1;
    pthread_mutex_init(&w->mu, NULL);
    w->xsize = xsize;
    w->ysize = ysize;
    w->across = (xsize + TILE - 1) / TILE;
    w->down = (ysize + TILE - 1) / TILE;
    memcpy(w->gt, gt, sizeof w->gt);
    nt = (size_t)w->across * (size_t)w->down;
    w->offsets = calloc(nt, sizeof *w->offsets);
    w->counts = calloc(nt, sizeof *w->counts);
    w->path = strdup(path);
    w->part = malloc(strlen(path) + 6);
    if (w->part)
        sprintf(w->part, "%s.part", path);
    if (!w->offsets || !w->counts || !w->path || !w->part || copy_georef(&w->georef, georef) != 0)
        goto oom;
    /* an existing raster of that name stays as it is until this one is complete */
    w->fd = open(w->part, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (w->fd < 0 || write_all(w->fd, header, sizeof header, 0) != 0) {
        snprintf(err, errcap, "write error: cannot create %s: %s", path, strerror(errno));
        gcn10_tiff_abort(w);
        return NULL;
    }
    w->pos = sizeof header;
    return w;

oom:
    snprintf(err, errcap, "out of memory creating %s", path);
    if (w)
        gcn10_tiff_abort(w);
    return NULL;
}

int gcn10_tiff_tiles_across(const gcn10_tiff_writer *w)
{
    return w->across;
}

int gcn10_tiff_tiles_down(const gcn10_tiff_writer *w)
{
    return w->down;
}

int gcn10_tiff_put_tile(gcn10_tiff_writer *w, int tx, int ty, const void *zdata, size_t nbytes)
{
    int rc = 0;
    size_t idx;

    if (tx < 0 || ty < 0 || tx >= w->across || ty >= w->down || nbytes == 0)
        return -1;
    idx = (size_t)ty * (size_t)w->across + (size_t)tx;
    pthread_mutex_lock(&w->mu);
    if (w->pos + nbytes + (1u << 20) > 0xffffffffull) {
        w->failed = true;           /* classic TIFF offsets are 32 bit */
        rc = -1;
    }
    else if (write_all(w->fd, zdata, nbytes, w->pos) != 0) {
        w->failed = true;
        rc = -1;
    }
    else {
        w->offsets[idx] = (uint32_t)w->pos;
        w->counts[idx] = (uint32_t)nbytes;
        w->pos += nbytes;
    }
    pthread_mutex_unlock(&w->mu);
    return rc;
}

/* Many tiles of one raster at once (a strip's worth from the GPU encoder): one pwritev per
 * 1024 tiles instead of a pwrite per tile -- a block has 358 000 of them. */
int gcn10_tiff_put_tiles(gcn10_tiff_writer *w, int n, const int *tx, const int *ty, const void *const *zdata,
                         const uint32_t *nbytes)
{
    struct iovec iov[512];
    int rc = 0;

    pthread_mutex_lock(&w->mu);
    for (int i = 0; i < n && rc == 0;) {
        int m = 0;
        uint64_t total = 0, at = w->pos;

        for (; i + m < n && m < 512; m++) {
            if (tx[i + m] < 0 || ty[i + m] < 0 || tx[i + m] >= w->across || ty[i + m] >= w->down ||
                nbytes[i + m] == 0) {
                rc = -1;
                break;
            }
            iov[m].iov_base = (void *)zdata[i + m];
            iov[m].iov_len = nbytes[i + m];
            total += nbytes[i + m];
        }
        if (rc != 0)
            break;
        if (w->pos + total + (1u << 20) > 0xffffffffull) {
            rc = -1;                    /* classic TIFF offsets are 32 bit */
            break;
        }
        for (int k = 0; k < m;) {       /* pwritev may stop short */
            ssize_t got = pwritev(w->fd, iov + k, m - k, (off_t)at);

            if (got < 0 && errno == EINTR)
                continue;
            if (got <= 0) {
                rc = -1;
                break;
            }
            at += (uint64_t)got;
            while (k < m && (size_t)got >= iov[k].iov_len)
                got -= (ssize_t)iov[k++].iov_len;
            if (k < m && got > 0) {
                iov[k].iov_base = (char *)iov[k].iov_base + got;
                iov[k].iov_len -= (size_t)got;
            }
        }
        if (rc != 0)
            break;
        for (int k = 0; k < m; k++) {
            size_t idx = (size_t)ty[i + k] * (size_t)w->across + (size_t)tx[i + k];

            w->offsets[idx] = (uint32_t)w->pos;
            w->counts[idx] = nbytes[i + k];
            w->pos += nbytes[i + k];
        }
        i += m;
    }
    if (rc != 0)
        w->failed = true;
    pthread_mutex_unlock(&w->mu);
    return rc;
}

/* O_DIRECT for the tile data of this file (the directory is written without it at the end).  Extents
 * handed to gcn10_tiff_put_extent must then start at 4096-aligned addresses and be readable up to the
 * next multiple of 4096 past their end -- the pinned copy of the GPU encoder's arena is
 * (DIRECT_ALIGN = the encoder's "arena_segment_align").  A file system that refuses the flag leaves the
 * writer as it was: returns 0 when direct I/O is on, -1 when it is not. */
enum { DIRECT_ALIGN = 4096 };

int gcn10_tiff_set_direct(gcn10_tiff_writer *w, bool on)
{
    int fl = fcntl(w->fd, F_GETFL);

    if (fl < 0)
        return -1;
    if (on == w->direct)
        return on ? 0 : -1;
#ifdef O_DIRECT
    if (fcntl(w->fd, F_SETFL, on ? (fl | O_DIRECT) : (fl & ~O_DIRECT)) != 0)
        return -1;
    w->direct = on;
    return on ? 0 : -1;
#else
    return -1;
#endif
}

/* n tiles of the raster whose streams lie in ONE extent of memory, `extent_bytes` long, stream i at
 * rel_off[i] (the layout the GPU encoders produce: a raster's streams of a strip back to back in tile order):
 * one write for all of them.  Same file contents as n gcn10_tiff_put_tile calls except for the few
 * alignment bytes between streams, which no directory entry points at. */
int gcn10_tiff_put_extent(gcn10_tiff_writer *w, const void *data, size_t extent_bytes, int n, const int *tx,
                          const int *ty, const uint32_t *rel_off, const uint32_t *nbytes)
{
    int rc = 0;

    for (int i = 0; i < n; i++)
        if (tx[i] < 0 || ty[i] < 0 || tx[i] >= w->across || ty[i] >= w->down || nbytes[i] == 0 ||
            (size_t)rel_off[i] + nbytes[i] > extent_bytes)
            return -1;
    pthread_mutex_lock(&w->mu);
    {
        uint64_t at = w->pos;
        size_t len = extent_bytes;

        if (w->direct) {
            at = (at + DIRECT_ALIGN - 1) / DIRECT_ALIGN * DIRECT_ALIGN;
            len = (len + DIRECT_ALIGN - 1) / DIRECT_ALIGN * DIRECT_ALIGN;
        }
        if (at + len + (1u << 20) > 0xffffffffull) {
            rc = -1;                    /* classic TIFF offsets are 32 bit */
        }
        else if (write_all(w->fd, data, len, at) != 0) {
            if (w->direct && errno == EINVAL && gcn10_tiff_set_direct(w, false) != 0 && !w->direct) {
                /* this file system takes the flag and then refuses the write: once more without it */
                at = w->pos;
                len = extent_bytes;
                rc = write_all(w->fd, data, len, at) != 0 ? -1 : 0;
            }
            else {
                rc = -1;
            }
        }
        if (rc == 0) {
            for (int i = 0; i < n; i++) {
                size_t idx = (size_t)ty[i] * (size_t)w->across + (size_t)tx[i];

                w->offsets[idx] = (uint32_t)(at + rel_off[i]);
                w->counts[idx] = nbytes[i];
            }
            w->pos = at + len;
        }
    }
    if (rc != 0)
        w->failed = true;
    pthread_mutex_unlock(&w->mu);
    return rc;
}

static void put16(unsigned char *p, unsigned v)
{
    p[0] = (unsigned char)(v & 0xff);
    p[1] = (unsigned char)(v >> 8);
}

static void put32(unsigned char *p, uint32_t v)
{
    p[0] = (unsigned char)(v & 0xff);
    p[1] = (unsigned char)((v >> 8) & 0xff);
    p[2] = (unsigned char)((v >> 16) & 0xff);
    p[3] = (unsigned char)(v >> 24);
}

struct dirent_w {
    unsigned tag, type;
    uint32_t count;
    const void *data;       /* little-endian payload */
    size_t nbytes;
};

int gcn10_tiff_finish(gcn10_tiff_writer *w, char *err, size_t errcap)
{
    size_t nt = (size_t)w->across * (size_t)w->down;
    double scale[3] = { w->gt[1], -w->gt[5], 0.0 };
    double tie[6] = { 0, 0, 0, w->gt[0], w->gt[3], 0 };
    double xform[16] = { 0 };
    unsigned char s_w[4], s_h[4], s_bps[2], s_comp[2], s_phot[2], s_spp[2], s_plan[2], s_tw[2],
        s_th[2], s_fmt[2];
    struct dirent_w ents[20];
    int ne = 0, rc = -1;
    unsigned char *dir = NULL;
    uint64_t data_pos, dir_pos;
    size_t dirbytes;

    if (w->failed) {
        snprintf(err, errcap, "write error on %s", w->path);
        goto done;
    }
    if (w->direct)
        gcn10_tiff_set_direct(w, false);        /* the directory is not sector sized */
    for (size_t i = 0; i < nt; i++)
        if (w->counts[i] == 0) {
            snprintf(err, errcap, "write error on %s: tile %zu was never written", w->path, i);
            goto done;
        }
    put32(s_w, (uint32_t)w->xsize);
    put32(s_h, (uint32_t)w->ysize);
    put16(s_bps, 8);
    put16(s_comp, 8);           /* COMPRESS=DEFLATE -> Adobe deflate */
    put16(s_phot, 1);           /* MinIsBlack */
    put16(s_spp, 1);
    put16(s_plan, 1);
    put16(s_tw, TILE);
    put16(s_th, TILE);
    put16(s_fmt, 1);            /* unsigned integer */
#define ENT(tag_, type_, count_, data_, nbytes_) \
    ents[ne++] = (struct dirent_w){ tag_, type_, (uint32_t)(count_), data_, nbytes_ }
    ENT(256, T_LONG, 1, s_w, 4);
    ENT(257, T_LONG, 1, s_h, 4);
    ENT(258, T_SHORT, 1, s_bps, 2);
    ENT(259, T_SHORT, 1, s_comp, 2);
    ENT(262, T_SHORT, 1, s_phot, 2);
    ENT(277, T_SHORT, 1, s_spp, 2);
    ENT(284, T_SHORT, 1, s_plan, 2);
    ENT(322, T_SHORT, 1, s_tw, 2);
    ENT(323, T_SHORT, 1, s_th, 2);
    ENT(324, T_LONG, nt, w->offsets, nt * 4);      /* host is little endian (x86-64) */
    ENT(325, T_LONG, nt, w->counts, nt * 4);
    ENT(339, T_SHORT, 1, s_fmt, 2);
    if (w->gt[2] == 0.0 && w->gt[4] == 0.0) {
        ENT(33550, T_DOUBLE, 3, scale, sizeof scale);
        ENT(33922, T_DOUBLE, 6, tie, sizeof tie);
    }
    else {
        /* a rotated or sheared geotransform: ModelTransformationTag, the 4x4 matrix GDAL's
         * GTiff driver writes for GDALSetGeoTransform in that case (src/raster.c:210) */
        xform[0] = w->gt[1]; xform[1] = w->gt[2]; xform[3] = w->gt[0];
        xform[4] = w->gt[4]; xform[5] = w->gt[5]; xform[7] = w->gt[3];
        xform[15] = 1.0;
        ENT(34264, T_DOUBLE, 16, xform, sizeof xform);
    }
    if (w->georef.n_geokeys > 0 && w->georef.geokeys)
        ENT(34735, T_SHORT, w->georef.n_geokeys, w->georef.geokeys, (size_t)w->georef.n_geokeys * 2);
    if (w->georef.geodoubles)
        ENT(34736, T_DOUBLE, w->georef.n_geodoubles, w->georef.geodoubles,
            (size_t)w->georef.n_geodoubles * 8);
    if (w->georef.geoascii)
        ENT(34737, T_ASCII, strlen(w->georef.geoascii) + 1, w->georef.geoascii,
            strlen(w->georef.geoascii) + 1);
#undef ENT

    /* out-of-line payloads first, then the directory, both word aligned */
    data_pos = (w->pos + 1) & ~1ull;
    dirbytes = 2 + (size_t)ne * 12 + 4;
    dir = calloc(1, dirbytes);
    if (!dir) {
        snprintf(err, errcap, "out of memory finishing %s", w->path);
        goto done;
    }
    put16(dir, (unsigned)ne);
    for (int i = 0; i < ne; i++) {
        unsigned char *e = dir + 2 + i * 12;

        put16(e, ents[i].tag);
        put16(e + 2, ents[i].type);
        put32(e + 4, ents[i].count);
        if (ents[i].nbytes <= 4) {
            memcpy(e + 8, ents[i].data, ents[i].nbytes);
        }
        else {
            if (data_pos + ents[i].nbytes > 0xffffffffull ||
                write_all(w->fd, ents[i].data, ents[i].nbytes, data_pos) != 0) {
                snprintf(err, errcap, "write error on %s", w->path);
                goto done;
            }
            put32(e + 8, (uint32_t)data_pos);
            data_pos = (data_pos + ents[i].nbytes + 1) & ~1ull;
        }
    }
    dir_pos = data_pos;
    {
        unsigned char off[4];

        put32(off, (uint32_t)dir_pos);
        if (dir_pos + dirbytes > 0xffffffffull || write_all(w->fd, dir, dirbytes, dir_pos) != 0 ||
            write_all(w->fd, off, 4, 4) != 0) {
            snprintf(err, errcap, "write error on %s", w->path);
            goto done;
        }
    }
    rc = 0;
done:
    free(dir);
    if (w->fd >= 0 && close(w->fd) != 0 && rc == 0) {
        snprintf(err, errcap, "write error closing %s: %s", w->path, strerror(errno));
        rc = -1;
    }
    w->fd = -1;
    if (rc == 0 && rename(w->part, w->path) != 0) {
        snprintf(err, errcap, "write error on %s: %s", w->path, strerror(errno));
        rc = -1;
    }
    if (rc == 0) {
        free(w->part);
        w->part = NULL;             /* nothing left for abort to remove */
    }
    gcn10_tiff_abort(w);
    return rc;
}

void gcn10_tiff_abort(gcn10_tiff_writer *w)
{
    if (!w)
        return;
    if (w->fd >= 0)
        close(w->fd);
    if (w->part)
        unlink(w->part);            /* an unfinished raster is no raster */
    free(w->part);
    free(w->offsets);
    free(w->counts);
    free(w->path);
    free_georef(&w->georef);
    pthread_mutex_destroy(&w->mu);
    free(w);
}

size_t gcn10_deflate_tile(const uint8_t *src, size_t stride, int valid_w, int valid_h, int level,
                          uint8_t *dst, size_t dstcap)
{
    unsigned char tile[TILE * TILE];
    uLongf dl = (uLongf)dstcap;

    if (valid_w <= 0 || valid_h <= 0 || valid_w > TILE || valid_h > TILE)
        return 0;
    if (valid_w < TILE || valid_h < TILE)
        memset(tile, 0, sizeof tile);
    for (int y = 0; y < valid_h; y++)
        memcpy(tile + (size_t)y * TILE, src + (size_t)y * stride, (size_t)valid_w);
    if (compress2(dst, &dl, tile, sizeof tile, level > 0 ? level : Z_DEFAULT_COMPRESSION) != Z_OK)
        return 0;
    return (size_t)dl;
}

int gcn10_save_raster(const uint8_t *data, int xsize, int ysize, const double gt[6],
                      const gcn10_georef *georef, const char *path, int level, char *err,
                      size_t errcap)
{
    gcn10_tiff_writer *w = gcn10_tiff_create(path, xsize, ysize, gt, georef, err, errcap);
    size_t cap = compressBound(TILE * TILE);
    uint8_t *z;

    if (!w)
        return -1;
    z = malloc(cap);
    if (!z) {
        snprintf(err, errcap, "out of memory writing %s", path);
        gcn10_tiff_abort(w);
        return -1;
    }
    for (int ty = 0; ty < w->down; ty++) {
        for (int tx = 0; tx < w->across; tx++) {
            int vw = xsize - tx * TILE < TILE ? xsize - tx * TILE : TILE;
            int vh = ysize - ty * TILE < TILE ? ysize - ty * TILE : TILE;
            size_t n = gcn10_deflate_tile(data + (size_t)ty * TILE * (size_t)xsize + (size_t)tx * TILE,
                                          (size_t)xsize, vw, vh, level, z, cap);

            if (n == 0 || gcn10_tiff_put_tile(w, tx, ty, z, n) != 0) {
                snprintf(err, errcap, "write error %d on %s", 3, path);     /* CE_Failure, src/raster.c:221 */
                free(z);
                gcn10_tiff_abort(w);
                return -1;
            }
        }
    }
    free(z);
    return gcn10_tiff_finish(w, err, errcap);
}

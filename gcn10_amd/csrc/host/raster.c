/* raster.c -- raster inputs (GeoTIFF or VRT mosaic) behind one handle.
 *
 * load_raster() of the reference (/root/reference/src/raster.c:106-189) is
 * GDALOpen + window arithmetic + one GDALRasterIO.  Here: gcn10_raster_open +
 * gcn10_raster_window (geo.c, the bit-exact arithmetic) + gcn10_raster_read,
 * which may be called strip by strip so the window can be staged through a
 * small pinned buffer instead of one whole-block malloc.
 *
 * VRT support covers what landcover/esa_worldcover_2021.vrt uses: one Byte band
 * of Simple/ComplexSource elements with equal-size SrcRect/DstRect and an
 * optional <NODATA>.  Its sources are /vsicurl/ URLs; offline they resolve to
 * "<esa_tile_dir>/<file name>" when the config names a local mirror.
 */
#include "gcn10_host.h"
#include "host_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct vrt_source {
    char *path;
    int sx, sy;             /* SrcRect offset */
    int dx, dy, w, h;       /* DstRect */
    int nodata;             /* -1 = none (SimpleSource) */
};

struct gcn10_raster {
    struct gcn10_tiff *tiff;        /* plain TIFF, or NULL for a VRT */
    int xsize, ysize;
    double gt[6];
    struct vrt_source *src;
    int n_src;
    gcn10_georef georef;            /* VRT: empty = WGS84 default in the writer */
};

static char *slurp_text(const char *path)
{
    FILE *f = fopen(path, "rb");
    char *buf;
    long n;

    if (!f)
        return NULL;
    if (fseek(f, 0, SEEK_END) != 0 || (n = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        return NULL;
    }
    buf = malloc((size_t)n + 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
        free(buf);
        fclose(f);
        return NULL;
    }
    fclose(f);
    buf[n] = '\0';
    return buf;
}

/* value of attribute `name` inside the tag text [tag, tag_end) */
static int attr_int(const char *tag, const char *tag_end, const char *name, int *out)
{
    size_t nl = strlen(name);

    for (const char *p = tag; p + nl + 2 < tag_end; p++) {
        if ((p == tag || p[-1] == ' ' || p[-1] == '\t' || p[-1] == '\n') &&
            strncmp(p, name, nl) == 0 && p[nl] == '=' && p[nl + 1] == '"') {
            *out = (int)strtod(p + nl + 2, NULL);
            return 0;
        }
    }
    return -1;
}

static char *dirname_dup(const char *path)
{
    const char *slash = strrchr(path, '/');
    size_t n = slash ? (size_t)(slash - path) : 0;
    char *d = malloc(n + 2);

    if (!d)
        return NULL;
    if (n == 0) {
        strcpy(d, slash ? "/" : ".");
    }
    else {
        memcpy(d, path, n);
        d[n] = '\0';
    }
    return d;
}

static int parse_vrt(struct gcn10_raster *r, const char *path, const char *tile_dir, char *err,
                     size_t errcap)
{
    char *xml = slurp_text(path);
    char *vdir = NULL;
    const char *p, *ds;
    int cap = 0;

    if (!xml) {
        snprintf(err, errcap, "gdal open failed: %s", path);
        return -1;
    }
    ds = strstr(xml, "<VRTDataset");
    if (!ds) {
        snprintf(err, errcap, "gdal open failed: %s (no VRTDataset element)", path);
        goto fail;
    }
    {
        const char *end = strchr(ds, '>');

        if (!end || attr_int(ds, end, "rasterXSize", &r->xsize) != 0 ||
            attr_int(ds, end, "rasterYSize", &r->ysize) != 0) {
            snprintf(err, errcap, "gdal open failed: %s (raster size missing)", path);
            goto fail;
        }
    }
    r->gt[0] = 0; r->gt[1] = 1; r->gt[2] = 0; r->gt[3] = 0; r->gt[4] = 0; r->gt[5] = 1;
    p = strstr(xml, "<GeoTransform>");
    if (p) {
        const char *q = p + strlen("<GeoTransform>");

        for (int i = 0; i < 6; i++) {
            char *endp;

            r->gt[i] = strtod(q, &endp);    /* same decimal->double conversion GDAL's CPLAtof gives */
            q = endp;
            while (*q == ',' || *q == ' ' || *q == '\n' || *q == '\t')
                q++;
        }
    }
    vdir = dirname_dup(path);
    if (!vdir)
        goto oom;

    p = xml;
    for (;;) {
        const char *a = strstr(p, "<ComplexSource");
        const char *b = strstr(p, "<SimpleSource");
        const char *s = (a && (!b || a < b)) ? a : b;
        const char *close_tag, *e, *fn, *fn_txt, *fn_end, *rect, *rect_end;
        bool complex_src;
        struct vrt_source src;
        int relative = 0;

        if (!s)
            break;
        complex_src = s == a;
        close_tag = complex_src ? "</ComplexSource>" : "</SimpleSource>";
        e = strstr(s, close_tag);
        if (!e)
            break;
        memset(&src, 0, sizeof src);
        src.nodata = -1;

        fn = strstr(s, "<SourceFilename");
        if (!fn || fn > e)
            goto next;
        fn_txt = strchr(fn, '>');
        if (!fn_txt || fn_txt > e)
            goto next;
        attr_int(fn, fn_txt, "relativeToVRT", &relative);
        fn_txt++;
        fn_end = strstr(fn_txt, "</SourceFilename>");
        if (!fn_end || fn_end > e)
            goto next;
        {
            size_t n = (size_t)(fn_end - fn_txt);
            char *name = malloc(n + 1);
            const char *base;

            if (!name)
                goto oom;
            memcpy(name, fn_txt, n);
            name[n] = '\0';
            base = strrchr(name, '/');
            base = base ? base + 1 : name;
            if (strncmp(name, "/vsi", 4) == 0) {
                /* /vsicurl/https://... : not reachable offline; use the local mirror */
                if (tile_dir && *tile_dir) {
                    src.path = malloc(strlen(tile_dir) + strlen(base) + 2);
                    if (src.path)
                        sprintf(src.path, "%s/%s", tile_dir, base);
                }
                else {
                    src.path = strdup(name);    /* opening it fails with a clear message */
                }
            }
            else if (relative && name[0] != '/') {
                src.path = malloc(strlen(vdir) + n + 2);
                if (src.path)
                    sprintf(src.path, "%s/%s", vdir, name);
            }
            else {
                src.path = strdup(name);
            }
            free(name);
            if (!src.path)
                goto oom;
        }
        rect = strstr(s, "<SrcRect");
        rect_end = rect ? strchr(rect, '>') : NULL;
        if (rect && rect < e && rect_end) {
            int sw = 0, sh = 0;

            attr_int(rect, rect_end, "xOff", &src.sx);
            attr_int(rect, rect_end, "yOff", &src.sy);
            attr_int(rect, rect_end, "xSize", &sw);
            attr_int(rect, rect_end, "ySize", &sh);
            src.w = sw;
            src.h = sh;
        }
        rect = strstr(s, "<DstRect");
        rect_end = rect ? strchr(rect, '>') : NULL;
        if (rect && rect < e && rect_end) {
            int dw = 0, dh = 0;

            attr_int(rect, rect_end, "xOff", &src.dx);
            attr_int(rect, rect_end, "yOff", &src.dy);
            attr_int(rect, rect_end, "xSize", &dw);
            attr_int(rect, rect_end, "ySize", &dh);
            if ((src.w && dw != src.w) || (src.h && dh != src.h)) {
                snprintf(err, errcap, "gdal open failed: %s (resampling VRT sources are not supported)", path);
                free(src.path);
                goto fail;
            }
            src.w = dw;
            src.h = dh;
        }
        if (complex_src) {
            const char *nd = strstr(s, "<NODATA>");

            if (nd && nd < e)
                src.nodata = (int)strtod(nd + 8, NULL);
        }
        if (src.w > 0 && src.h > 0) {
            if (r->n_src == cap) {
                struct vrt_source *g = realloc(r->src, (size_t)(cap ? cap * 2 : 64) * sizeof *g);

                if (!g) {
                    free(src.path);
                    goto oom;
                }
                r->src = g;
                cap = cap ? cap * 2 : 64;
            }
            r->src[r->n_src++] = src;
        }
        else {
            free(src.path);
        }
next:
        p = e + 1;
    }
    free(vdir);
    free(xml);
    return 0;

oom:
    snprintf(err, errcap, "out of memory for raster %s", path);
fail:
    free(vdir);
    free(xml);
    return -1;
}

gcn10_raster *gcn10_raster_open(const char *path, const char *vrt_tile_dir, char *err, size_t errcap)
{
    gcn10_raster *r = calloc(1, sizeof *r);
    size_t n = strlen(path);

    if (!r) {
        snprintf(err, errcap, "out of memory for raster %s", path);
        return NULL;
    }
    if (n > 4 && strcasecmp(path + n - 4, ".vrt") == 0) {
        if (parse_vrt(r, path, vrt_tile_dir, err, errcap) != 0) {
            gcn10_raster_close(r);
            return NULL;
        }
        return r;
    }
    r->tiff = gcn10_tiff_open_reader(path, err, errcap);
    if (!r->tiff) {
        gcn10_raster_close(r);
        return NULL;
    }
    gcn10_tiff_reader_info(r->tiff, &r->xsize, &r->ysize, r->gt);
    return r;
}

void gcn10_raster_close(gcn10_raster *r)
{
    if (!r)
        return;
    gcn10_tiff_close_reader(r->tiff);
    for (int i = 0; i < r->n_src; i++)
        free(r->src[i].path);
    free(r->src);
    free(r);
}

void gcn10_raster_info(const gcn10_raster *r, int *xsize, int *ysize, double gt[6])
{
    *xsize = r->xsize;
    *ysize = r->ysize;
    memcpy(gt, r->gt, sizeof r->gt);
}

const gcn10_georef *gcn10_raster_georef(const gcn10_raster *r)
{
    return r->tiff ? gcn10_tiff_reader_georef(r->tiff) : &r->georef;
}

int gcn10_raster_read(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount, uint8_t *dst,
                      char *err, size_t errcap)
{
    return gcn10_raster_read_mt(r, xoff, yoff, xcount, ycount, dst, NULL, err, errcap);
}

int gcn10_raster_read_mt(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount, uint8_t *dst,
                         gcn10_pool *pool, char *err, size_t errcap)
{
    if (xoff < 0 || yoff < 0 || xcount <= 0 || ycount <= 0 || xoff + xcount > r->xsize ||
        yoff + ycount > r->ysize) {
        snprintf(err, errcap, "window %d,%d %dx%d outside raster %dx%d", xoff, yoff, xcount, ycount,
                 r->xsize, r->ysize);
        return -1;
    }
    if (r->tiff)
        return gcn10_tiff_read_window_mt(r->tiff, xoff, yoff, xcount, ycount, dst, (size_t)xcount, pool,
                                         err, errcap);

    /* VRT: start from 0 (the band's NoDataValue in the shipped VRT) and paint
     * the sources in file order */
    memset(dst, 0, (size_t)xcount * (size_t)ycount);
    for (int i = 0; i < r->n_src; i++) {
        const struct vrt_source *s = &r->src[i];
        int x0 = xoff > s->dx ? xoff : s->dx;
        int y0 = yoff > s->dy ? yoff : s->dy;
        int x1 = xoff + xcount < s->dx + s->w ? xoff + xcount : s->dx + s->w;
        int y1 = yoff + ycount < s->dy + s->h ? yoff + ycount : s->dy + s->h;
        struct gcn10_tiff *t;
        uint8_t *at;
        int rc;

        if (x0 >= x1 || y0 >= y1)
            continue;
        t = gcn10_tiff_open_reader(s->path, err, errcap);
        if (!t)
            return -1;
        at = dst + (size_t)(y0 - yoff) * (size_t)xcount + (size_t)(x0 - xoff);
        if (s->nodata < 0) {
            rc = gcn10_tiff_read_window_mt(t, s->sx + (x0 - s->dx), s->sy + (y0 - s->dy), x1 - x0,
                                           y1 - y0, at, (size_t)xcount, pool, err, errcap);
        }
        else {
            /* ComplexSource with NODATA: source pixels equal to it stay transparent */
            size_t w = (size_t)(x1 - x0), h = (size_t)(y1 - y0);
            uint8_t *tmp = malloc(w * h);

            if (!tmp) {
                snprintf(err, errcap, "out of memory for raster %s", s->path);
                rc = -1;
            }
            else {
                rc = gcn10_tiff_read_window_mt(t, s->sx + (x0 - s->dx), s->sy + (y0 - s->dy), x1 - x0,
                                               y1 - y0, tmp, w, pool, err, errcap);
                if (rc == 0)
                    for (size_t y = 0; y < h; y++)
                        for (size_t x = 0; x < w; x++)
                            if (tmp[y * w + x] != (uint8_t)s->nodata)
                                at[y * (size_t)xcount + x] = tmp[y * w + x];
                free(tmp);
            }
        }
        gcn10_tiff_close_reader(t);
        if (rc != 0)
            return -1;
    }
    return 0;
}

void gcn10_read_plan_free(struct gcn10_read_plan *plan)
{
    for (int i = 0; i < plan->n_opened; i++)
        gcn10_tiff_close_reader(plan->opened[i]);
    free(plan->opened);
    free(plan->chunks);
    memset(plan, 0, sizeof *plan);
}

int gcn10_raster_plan_window(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount,
                             struct gcn10_read_plan *plan, char *err, size_t errcap)
{
    int rc = 0;

    memset(plan, 0, sizeof *plan);
    if (xoff < 0 || yoff < 0 || xcount <= 0 || ycount <= 0 || xoff + xcount > r->xsize ||
        yoff + ycount > r->ysize) {
        snprintf(err, errcap, "window %d,%d %dx%d outside raster %dx%d", xoff, yoff, xcount, ycount,
                 r->xsize, r->ysize);
        return -1;
    }
    if (r->tiff) {
        rc = gcn10_tiff_plan_window(r->tiff, xoff, yoff, xcount, ycount, 0, 0, plan, err, errcap);
        if (rc != 0)
            gcn10_read_plan_free(plan);
        return rc;
    }
    /* VRT: every source that touches the window.  The chunks of a plan are written
     * concurrently, so sources that overlap inside the window (painted in file order by the
     * host reader) are left to it; NODATA other than the 0 background too. */
    for (int i = 0; i < r->n_src && rc == 0; i++) {
        const struct vrt_source *s = &r->src[i];
        int x0 = xoff > s->dx ? xoff : s->dx;
        int y0 = yoff > s->dy ? yoff : s->dy;
        int x1 = xoff + xcount < s->dx + s->w ? xoff + xcount : s->dx + s->w;
        int y1 = yoff + ycount < s->dy + s->h ? yoff + ycount : s->dy + s->h;
        struct gcn10_tiff *t, **g;

        if (x0 >= x1 || y0 >= y1)
            continue;
        if (s->nodata > 0) {
            rc = 1;
            break;
        }
        for (int k = 0; k < i; k++) {
            const struct vrt_source *o = &r->src[k];
            int ox0 = x0 > o->dx ? x0 : o->dx, oy0 = y0 > o->dy ? y0 : o->dy;
            int ox1 = x1 < o->dx + o->w ? x1 : o->dx + o->w, oy1 = y1 < o->dy + o->h ? y1 : o->dy + o->h;

            if (ox0 < ox1 && oy0 < oy1)
                rc = 1;
        }
        if (rc != 0)
            break;
        t = gcn10_tiff_open_reader(s->path, err, errcap);
        if (!t) {
            rc = -1;
            break;
        }
        g = realloc(plan->opened, (size_t)(plan->n_opened + 1) * sizeof *g);
        if (!g) {
            gcn10_tiff_close_reader(t);
            snprintf(err, errcap, "out of memory for the read plan");
            rc = -1;
            break;
        }
        plan->opened = g;
        plan->opened[plan->n_opened++] = t;
        rc = gcn10_tiff_plan_window(t, s->sx + (x0 - s->dx), s->sy + (y0 - s->dy), x1 - x0, y1 - y0,
                                    x0 - xoff, y0 - yoff, plan, err, errcap);
    }
    if (rc != 0)
        gcn10_read_plan_free(plan);
    return rc;
}

/* pipeline.c -- the run: block queue over GPUs and the per-block pipeline.
 *
 * What the reference does per block on one MPI rank
 * (/root/reference/src/cn.c:134-384): find the block's bbox, load the
 * landcover window and the soil window with GDAL, upsample, then 18 times
 * {read CSV, memcpy, remap, memset, lookup, GDAL DEFLATE write}.
 *
 * Here, per block, two threads per GPU worker:
 *   input  (pipeline_input.c, one block AHEAD): bbox -> windows (geo.c, bit-exact) -> soil window +
 *          index maps -> HBM; the landcover window's chunks as they lie in the files -> pinned ring ->
 *          HBM -> decoded / untiled there into the row-major block
 *   encode (here): per strip of rows ONE fused device pass turns landcover + soil into the compressed
 *          256x256 tiles of all 18 rasters (no CN raster in HBM); the tiles of a raster and a strip are
 *          one extent of the arena, copied back on the d2h stream into pinned memory; buffer sets rotate,
 *          so kernels, copy-back and file writes of neighbouring strips overlap
 *   sink   one write per raster and strip appends the extent to the raster's GeoTIFF (I/O pool)
 * Blocks are pulled from one atomic counter by all workers (the reference's
 * static `i += size` round-robin, src/main.c:171, made dynamic); no data moves
 * between GPUs, so there is no collective and no RCCL.
 */
#include "pipeline_internal.h"

#include <errno.h>
#include <limits.h>
#include <pthread.h>
#include <sched.h>
#include <stdarg.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

double gcn10_now_seconds(void)
{
    struct timespec ts;

    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void gcn10_wlog(struct worker *w, const char *level, bool console, const char *fmt, ...)
{
    char msg[8192];                         /* char msg[8192], src/cn.c:144 */
    va_list ap;

    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    gcn10_log_message(w->log, level, msg, console);
}

#define now_seconds gcn10_now_seconds
#define wlog gcn10_wlog

/* ------------------------------------------------------------------------ */
/* sink: tile compression jobs                                               */
/* ------------------------------------------------------------------------ */

struct tile_job {
    struct worker *w;
    struct strip_buf *b;
    gcn10_tiff_writer *tif;
    const uint8_t *strip;                   /* pinned strip of one raster */
    int W, H;                               /* raster size */
    int y0;                                 /* first raster row of the strip */
    int ty;                                 /* tile row (global) */
};

static void tile_row_job(void *arg)
{
    struct tile_job *j = arg;
    struct run *r = j->w->run;
    uint8_t z[TILE * TILE + TILE * TILE / 1000 + 64];       /* >= compressBound(65536) */
    int across = gcn10_tiff_tiles_across(j->tif);
    int row0 = j->ty * TILE;
    int vh = j->H - row0 < TILE ? j->H - row0 : TILE;

    for (int tx = 0; tx < across; tx++) {
        int vw = j->W - tx * TILE < TILE ? j->W - tx * TILE : TILE;
        const uint8_t *src = j->strip + (size_t)(row0 - j->y0) * (size_t)j->W + (size_t)tx * TILE;
        size_t n = gcn10_deflate_tile(src, (size_t)j->W, vw, vh, r->deflate_level, z, sizeof z);

        if (n == 0 || gcn10_tiff_put_tile(j->tif, tx, j->ty, z, n) != 0) {
            atomic_store(&j->w->failed, true);
            break;
        }
    }
    pthread_mutex_lock(&j->b->mu);
    if (--j->b->pending == 0)
        pthread_cond_broadcast(&j->b->cv);
    pthread_mutex_unlock(&j->b->mu);
    free(j);
}

/* GPU-deflate sink: append the already compressed tiles of one raster */
struct put_job {
    struct worker *w;
    struct strip_buf *b;
    gcn10_tiff_writer *tif;
    int raster, ty0, across, down;          /* tile rows ty0 .. ty0+down of the raster */
};

static void put_tiles_job(void *arg)
{
    struct put_job *j = arg;
    const uint32_t *tab = j->b->h_table + (size_t)j->raster * (size_t)j->across * (size_t)j->down * 2;
    const size_t n = (size_t)j->across * (size_t)j->down;
    const bool arrived = true;              /* (the strip's gate job has waited for the copy: strip_gate_job) */
    int *txs = malloc(n * sizeof *txs), *tys = malloc(n * sizeof *tys);
    const void **data = malloc(n * sizeof *data);
    uint32_t *sizes = malloc(n * sizeof *sizes), *rel = malloc(n * sizeof *rel);
    bool ok = arrived && txs && tys && data && sizes && rel;
    bool extent = true;             /* the streams lie back to back in tile order (16-byte slots) */
    uint32_t first = 0, end = 0;

    for (int ty = 0; ok && ty < j->down; ty++) {
        for (int tx = 0; tx < j->across; tx++) {
            const size_t i = (size_t)ty * j->across + tx;
            const uint32_t off = tab[i * 2], size = tab[i * 2 + 1];

            /* the host copy holds exactly the `used` bytes the encoder produced, not arena_cap */
            if (off == 0xffffffffu || size == 0 || (size_t)off + size > j->b->h_tiles_used) {
                ok = false;
                break;
            }
            txs[i] = tx;
            tys[i] = j->ty0 + ty;
            data[i] = j->b->h_tiles + off;
            sizes[i] = size;
            if (i == 0)
                first = off;
            else if (off != ((end + 15u) & ~15u))
                extent = false;
            rel[i] = off - first;
            end = off + size;
        }
    }
    /* The GPU encoders lay a raster's streams of a strip out as ONE extent in tile order (pass B'): one
     * write appends them all.  (An alias raster of the fused encoder -- its tiles ARE another raster's
     * streams -- points into that raster's extent: the same test holds when every tile is aliased alike;
     * a mixture falls back to gathered writes, 512 tiles per system call.) */
    if (ok && extent)
        ok = gcn10_tiff_put_extent(j->tif, j->b->h_tiles + first, (size_t)(end - first), (int)n, txs, tys, rel, sizes) == 0;
    else if (ok)
        ok = gcn10_tiff_put_tiles(j->tif, (int)n, txs, tys, data, sizes) == 0;
    if (!ok)
        atomic_store(&j->w->failed, true);
    free(txs);
    free(tys);
    free(data);
    free(sizes);
    free(rel);
    pthread_mutex_lock(&j->b->mu);
    if (--j->b->pending == 0)
        pthread_cond_broadcast(&j->b->cv);
    pthread_mutex_unlock(&j->b->mu);
    free(j);
}

/* the end of a raster's file: directory, close, rename */
struct finish_job {
    struct worker *w;
    struct strip_buf *b;
    gcn10_tiff_writer *tif;
};

static void finish_tiff_job(void *arg)
{
    struct finish_job *j = arg;
    char err[1024] = "";

    if (gcn10_tiff_finish(j->tif, err, sizeof err) != 0)
        wlog(j->w, "ERROR", true, "%s", err);           /* "write error %d on %s", src/raster.c:221 */
    pthread_mutex_lock(&j->b->mu);
    if (--j->b->pending == 0)
        pthread_cond_broadcast(&j->b->cv);
    pthread_mutex_unlock(&j->b->mu);
    free(j);
}

/* One job per strip waits for the strip's compressed bytes to arrive in pinned memory (ONE thread of the pool in
 * hipEventSynchronize instead of one per raster, round 3) and then hands the rasters' extents to the pool. */
struct gate_job {
    struct worker *w;
    struct strip_buf *b;
    gcn10_tiff_writer *tifs[GCN10_N_RASTERS];
    int ty0, across, down;
};

static void strip_gate_job(void *arg)
{
    struct gate_job *g = arg;
    struct run *r = g->w->run;
    const bool arrived = r->gpu->event_sync(g->w->ctx, g->b->ev_d2h) == 0;

    if (!arrived)
        atomic_store(&g->w->failed, true);
    for (int q = 0; arrived && q < r->n_sel; q++) {       /* stream q of the table = the q-th selected raster */
        struct put_job *j = malloc(sizeof *j);

        if (!j) {
            atomic_store(&g->w->failed, true);
            break;
        }
        *j = (struct put_job){ g->w, g->b, g->tifs[r->sel[q]], q, g->ty0, g->across, g->down };
        pthread_mutex_lock(&g->b->mu);
        g->b->pending++;
        pthread_mutex_unlock(&g->b->mu);
        gcn10_pool_submit(r->pool, put_tiles_job, j);
    }
    pthread_mutex_lock(&g->b->mu);
    if (--g->b->pending == 0)
        pthread_cond_broadcast(&g->b->cv);
    pthread_mutex_unlock(&g->b->mu);
    free(g);
}

static void wait_sink(struct strip_buf *b)
{
    double t0 = now_seconds();

    pthread_mutex_lock(&b->mu);
    while (b->pending > 0)
        pthread_cond_wait(&b->cv, &b->mu);
    pthread_mutex_unlock(&b->mu);
    if (b->owner)
        b->owner->t_sink_wait += now_seconds() - t0;
}

/* hands the finished strip in buffer b to the compression pool */
static int drain_strip_inner(struct worker *w, struct strip_buf *b, gcn10_tiff_writer *tifs[GCN10_N_RASTERS],
                             int W, int H);

static int drain_strip(struct worker *w, struct strip_buf *b, gcn10_tiff_writer *tifs[GCN10_N_RASTERS],
                       int W, int H)
{
    double t0 = now_seconds();
    int rc = drain_strip_inner(w, b, tifs, W, H);

    w->t_gpu_wait += now_seconds() - t0;
    return rc;
}

static int drain_strip_inner(struct worker *w, struct strip_buf *b, gcn10_tiff_writer *tifs[GCN10_N_RASTERS],
                             int W, int H)
{
    struct run *r = w->run;

    if (!b->d2h_issued)
        return 0;
    b->d2h_issued = false;
    if (r->gpu_deflate) {
        /* the sizes are known only now: fetch exactly the bytes the encoder produced */
        const struct gcn10_gpu_api *g = r->gpu;
        int across = (W + TILE - 1) / TILE, down = (b->rows + TILE - 1) / TILE;
        size_t used;

        if (g->event_sync(w->ctx, b->ev_meta) != 0)
            goto gpu_error;
        used = (size_t)*b->h_cursor;
        if (used > b->arena_cap) {
            wlog(w, "ERROR", true, "gpu deflate arena overflow (%zu > %zu)", used, b->arena_cap);
            return -1;
        }
        if (r->null_sink)
            return 0;
        {
            /* CN rasters compress to a few percent, so the pinned arena is an eighth of the
             * worst case; a strip of noise that needs more takes a pageable detour */
            uint8_t *dst = b->h_arena;

            free(b->h_spill);
            b->h_spill = NULL;
            if (used > b->h_arena_cap) {
                b->h_spill = aligned_alloc(4096, (used + 4095) & ~(size_t)4095);
                if (!b->h_spill) {
                    wlog(w, "ERROR", true, "malloc failed for %zu bytes of compressed tiles", used);
                    return -1;
                }
                dst = b->h_spill;
            }
            b->h_tiles = dst;
            b->h_tiles_used = used;
            /* the sink jobs wait for this copy themselves: the worker goes on to the next strip */
            /* (whole 4096-byte units: an O_DIRECT write of the last extent reads up to the next multiple;
             * both arenas are that much larger than `used` can get) */
            if (g->memcpy_d2h(w->ctx, dst, b->d_arena, (used + 4095) & ~(size_t)4095, w->s_d2h) != 0 ||
                g->event_record(w->ctx, b->ev_d2h, w->s_d2h) != 0)
                goto gpu_error;
        }
        {
            struct gate_job *j = malloc(sizeof *j);

            if (!j) {
                wlog(w, "ERROR", true, "malloc failed for tile job");
                return -1;
            }
            j->w = w;
            j->b = b;
            memcpy(j->tifs, tifs, sizeof j->tifs);
            j->ty0 = b->y0 / TILE;
            j->across = across;
            j->down = down;
            pthread_mutex_lock(&b->mu);
            b->pending++;
            pthread_mutex_unlock(&b->mu);
            gcn10_pool_submit(r->pool, strip_gate_job, j);
        }
        (void)H;
        return 0;
gpu_error:
        wlog(w, "ERROR", true, "gpu: %s", g->last_error());
        return -1;
    }
    if (r->gpu->event_sync(w->ctx, b->ev_d2h) != 0) {
        wlog(w, "ERROR", true, "gpu: %s", r->gpu->last_error());
        return -1;
    }
    if (r->null_sink)
        return 0;
    for (int ty = b->y0 / TILE; ty * TILE < b->y0 + b->rows; ty++) {
        for (int q = 0; q < r->n_sel; q++) {
            const int k = r->sel[q];
            struct tile_job *j = malloc(sizeof *j);

            if (!j) {
                wlog(w, "ERROR", true, "malloc failed for tile job");
                return -1;
            }
            *j = (struct tile_job){ w, b, tifs[k], b->h_out[k], W, H, b->y0, ty };
            pthread_mutex_lock(&b->mu);
            b->pending++;
            pthread_mutex_unlock(&b->mu);
            gcn10_pool_submit(r->pool, tile_row_job, j);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* worker resources                                                          */
/* ------------------------------------------------------------------------ */

#define GPU_TRY(w, call)                                                       \
    do {                                                                       \
        if ((call) != 0) {                                                     \
            wlog((w), "ERROR", true, "gpu: %s", (w)->run->gpu->last_error());  \
            return -1;                                                         \
        }                                                                      \
    } while (0)

static void free_strip_buffers(struct worker *w)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    for (int i = 0; i < w->run->nbuf; i++) {
        struct strip_buf *b = &w->buf[i];

        for (int k = 0; k < GCN10_N_RASTERS; k++) {
            if (b->h_out[k]) g->host_free(w->ctx, b->h_out[k]);
            if (b->d_out[k]) g->free(w->ctx, b->d_out[k]);
            b->h_out[k] = b->d_out[k] = NULL;
        }
        if (b->h_arena) g->host_free(w->ctx, b->h_arena);
        free(b->h_spill);
        b->h_spill = NULL;
        if (b->d_arena) g->free(w->ctx, b->d_arena);
        if (b->h_table) g->host_free(w->ctx, b->h_table);
        if (b->d_table) g->free(w->ctx, b->d_table);
        if (b->h_cursor) g->host_free(w->ctx, b->h_cursor);
        if (b->d_cursor) g->free(w->ctx, b->d_cursor);
        if (b->d_ptrs) g->free(w->ctx, (void *)b->d_ptrs);
        b->h_arena = b->d_arena = NULL;
        b->h_table = b->d_table = NULL;
        b->h_cursor = b->d_cursor = NULL;
        b->d_ptrs = NULL;
    }
    w->buf_px = 0;
    w->buf_tiles = 0;
}

/* strip buffers sized for blocks W wide: nbuf x (1 + 18) pinned + the same in HBM */
static int ensure_strip_buffers(struct worker *w, int W)
{
    const struct gcn10_gpu_api *g = w->run->gpu;
    size_t px = (size_t)W * (size_t)w->strip_rows;
    size_t n_tiles = (size_t)((W + TILE - 1) / TILE) * (size_t)(w->strip_rows / TILE);

    if (px <= w->buf_px && n_tiles <= w->buf_tiles)
        return 0;
    free_strip_buffers(w);
    for (int i = 0; i < w->run->nbuf; i++) {
        struct strip_buf *b = &w->buf[i];

        for (int q = 0; q < w->run->n_sel; q++) {
            const int k = w->run->sel[q];

            if (!w->run->gpu_deflate)
            {
                GPU_TRY(w, g->host_alloc(w->ctx, px, (void **)&b->h_out[k]));
                atomic_fetch_add(&w->run->pinned_bytes, (long long)px);
            }
            if (!w->fused)
                GPU_TRY(w, g->malloc(w->ctx, px, (void **)&b->d_out[k]));
        }
        if (w->run->gpu_deflate) {
            size_t tiles = n_tiles;

            b->arena_cap = g->deflate_arena_bound(W, w->strip_rows, GCN10_N_RASTERS);
            GPU_TRY(w, g->malloc(w->ctx, b->arena_cap + 4096, (void **)&b->d_arena));
            b->h_arena_cap = b->arena_cap / 8 > ((size_t)32 << 20) ? b->arena_cap / 8 : ((size_t)32 << 20);
            if (getenv("GCN10_PINNED_ARENA_BYTES"))        /* tests: force the spill path */
                b->h_arena_cap = (size_t)strtoull(getenv("GCN10_PINNED_ARENA_BYTES"), NULL, 10);
            if (b->h_arena_cap < 4096)
                b->h_arena_cap = 4096;
            if (b->h_arena_cap > b->arena_cap)
                b->h_arena_cap = b->arena_cap;
            b->h_arena_cap &= ~(size_t)4095;
            GPU_TRY(w, g->host_alloc(w->ctx, b->h_arena_cap + 4096, (void **)&b->h_arena));
            atomic_fetch_add(&w->run->pinned_bytes, (long long)(b->h_arena_cap + 4096 + tiles * GCN10_N_RASTERS * 8));
            GPU_TRY(w, g->malloc(w->ctx, tiles * GCN10_N_RASTERS * 8, (void **)&b->d_table));
            GPU_TRY(w, g->host_alloc(w->ctx, tiles * GCN10_N_RASTERS * 8, (void **)&b->h_table));
            GPU_TRY(w, g->malloc(w->ctx, 8, (void **)&b->d_cursor));
            GPU_TRY(w, g->host_alloc(w->ctx, 8, (void **)&b->h_cursor));
            if (!w->fused) {
                /* the per-raster encoder takes the selected strips, packed in raster order */
                uint8_t *packed[GCN10_N_RASTERS] = { 0 };

                for (int q = 0; q < w->run->n_sel; q++)
                    packed[q] = b->d_out[w->run->sel[q]];
                GPU_TRY(w, g->malloc(w->ctx, GCN10_N_RASTERS * sizeof(void *), (void **)&b->d_ptrs));
                GPU_TRY(w, g->memcpy_h2d(w->ctx, (void *)b->d_ptrs, packed, GCN10_N_RASTERS * sizeof(void *),
                                         w->s_kernel));
                GPU_TRY(w, g->stream_sync(w->ctx, w->s_kernel));
            }
        }
    }
    w->buf_px = px;
    w->buf_tiles = n_tiles;
    return 0;
}

int gcn10_ensure_pinned_on(struct worker *w, gcn10_gpu_ctx *ctx, void **p, size_t *cap, size_t need)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    if (need <= *cap)
        return 0;
    if (*p)
        g->host_free(ctx, *p);
    *p = NULL;
    *cap = 0;
    need += need / 8 + 4096;
    GPU_TRY(w, g->host_alloc(ctx, need, p));
    atomic_fetch_add(&w->run->pinned_bytes, (long long)need);
    *cap = need;
    return 0;
}

int gcn10_ensure_dev_on(struct worker *w, gcn10_gpu_ctx *ctx, void **p, size_t *cap, size_t need)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    if (need <= *cap)
        return 0;
    if (*p)
        GPU_TRY(w, g->free(ctx, *p));
    *p = NULL;
    *cap = 0;
    GPU_TRY(w, g->malloc(ctx, need, p));
    *cap = need;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* one block: src/cn.c:134-384                                               */
/* ------------------------------------------------------------------------ */

/* output path with the reference's non-overwrite rule: an existing file is left
 * alone and the new one gets a trailing underscore (src/cn.c:320-360) */
static void output_path(char *out, size_t cap, const char *cond, const char *hc, const char *arc,
                        int block_id, bool overwrite)
{
    snprintf(out, cap, "cn_rasters_%s/cn_%s_%s_%d.tif", cond, hc, arc, block_id);  /* src/cn.c:308 */
    if (!overwrite) {
        FILE *f = fopen(out, "r");

        if (f) {
            fclose(f);
            snprintf(out, cap, "cn_rasters_%s/cn_%s_%s_%d_.tif", cond, hc, arc, block_id); /* :341 */
        }
    }
}

/* output directories (src/cn.c:237-256) and the files of the run's rasters, for block `in` */
int gcn10_create_outputs(struct worker *w, struct block_in *in)
{
    struct run *r = w->run;
    char err[1024] = "";
    double t_mark = now_seconds();

    memset(in->tifs, 0, sizeof in->tifs);
    in->tifs_ok = false;
    if (r->null_sink) {
        in->tifs_ok = true;
        return 0;
    }
    for (int c = 0; c < 2; c++) {
        char dir[64];

        if (!(r->cond_mask & (1u << c)))
            continue;
        snprintf(dir, sizeof dir, "cn_rasters_%s", gcn10_conds[c]);
        if (mkdir(dir, 0755) != 0 && errno != EEXIST) {
            wlog(w, "ERROR", true, "failed to create output directory %s", dir);       /* src/cn.c:250 */
            return -1;
        }
    }
    for (int q = 0; q < r->n_sel; q++) {
        const int k = r->sel[q];
        char path[PATH_MAX];

        output_path(path, sizeof path, gcn10_conds[k / 9], gcn10_hcs[(k % 9) / 3], gcn10_arcs[k % 3],
                    in->block_id, r->opt.overwrite);
        in->tifs[k] = gcn10_tiff_create(path, in->W, in->H, in->gt, gcn10_raster_georef(w->esa), err, sizeof err);
        if (!in->tifs[k]) {
            wlog(w, "ERROR", true, "%s", err);          /* save_raster logs and goes on, src/raster.c:220-223 */
            gcn10_abort_outputs(in);
            w->t_create += now_seconds() - t_mark;
            return 1;
        }
        if (r->direct_io && r->gpu_deflate)
            gcn10_tiff_set_direct(in->tifs[k], true);   /* best effort: a file system that refuses writes buffered */
    }
    in->tifs_ok = true;
    w->t_create += now_seconds() - t_mark;
    return 0;
}

void gcn10_abort_outputs(struct block_in *in)
{
    for (int k = 0; k < GCN10_N_RASTERS; k++) {
        if (in->tifs[k])
            gcn10_tiff_abort(in->tifs[k]);
        in->tifs[k] = NULL;
    }
    in->tifs_ok = false;
}

/* The back half of process_block (src/cn.c:236-384) for a block whose input the input side has put on its
 * way to HBM.  Returns 0 (done, or skipped like the reference skips) or -1 for errors the
 * reference answers with MPI_Abort */
static int encode_block(struct worker *w, struct block_in *in)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    char err[1024] = "";
    const int W = in->W, H = in->H, block_id = in->block_id;
    gcn10_tiff_writer *tifs[GCN10_N_RASTERS] = { 0 };
    int rc = 0, n_strips;
    bool ok = false;
    double t_mark = now_seconds();

    if (getenv("GCN10_TEST_FAIL_BLOCK") && atoi(getenv("GCN10_TEST_FAIL_BLOCK")) == block_id) {
        /* test hook (tools/r03/run_rehearse_8gpu.sh): one worker meets an error of the kind the reference answers
         * with MPI_Abort -- the whole run must stop and exit with code 1 (src/main.c:175, src/cn.c:25, 216) */
        wlog(w, "ERROR", true, "malloc failed for block %d (GCN10_TEST_FAIL_BLOCK)", block_id);
        return -1;
    }
    /* the 18 files were created by the input side (gcn10_create_outputs), one block ahead */
    memcpy(tifs, in->tifs, sizeof tifs);
    memset(in->tifs, 0, sizeof in->tifs);
    if (!r->null_sink && !in->tifs_ok)
        goto out;                       /* save_raster logs and goes on, src/raster.c:220-223: logged at creation */
    t_mark = now_seconds();

    /* device side of the block: the encoder waits (on the device) for the input side's copies and kernels */
    w->strip_rows = r->strip_rows;
    if (ensure_strip_buffers(w, W) != 0) {
        rc = -1;
        goto out;
    }
    atomic_store(&w->failed, false);
    if (g->stream_wait_event(w->ctx, w->s_kernel, in->ev_ready) != 0 ||
        g->prepare_tile(w->ctx, in->d_coarse, in->hsx, in->hsy, in->d_ci, W, w->s_kernel) != 0) {
        wlog(w, "ERROR", true, "gpu: %s", g->last_error());
        rc = -1;
        goto out;
    }
    w->t_device += now_seconds() - t_mark;

    /* strips: rows are multiples of 256 (whole GeoTIFF tile rows) and of 16
     * (16-byte aligned strip starts for any W) */
    n_strips = (H + w->strip_rows - 1) / w->strip_rows;
    for (int s = 0; s < n_strips; s++) {
        struct strip_buf *b = &w->buf[s % r->nbuf];
        int y0 = s * w->strip_rows;
        int rows = H - y0 < w->strip_rows ? H - y0 : w->strip_rows;
        size_t px = (size_t)W * (size_t)rows;
        uint8_t *outs[GCN10_N_RASTERS];
        const uint8_t *d_esa = in->d_block + (size_t)y0 * (size_t)W;

        /* this buffer's previous strip: its D2H must be done and handed to the sink,
         * and the sink must be done with the pinned buffers */
        if (drain_strip(w, b, tifs, W, H) != 0) {
            rc = -1;
            goto out;
        }
        wait_sink(b);

        /* the strip's kernels on the compute stream */
        if (w->fused) {
            /* landcover + soil -> 18 x compressed tiles in one device pass, no CN strip in HBM */
            if (g->deflate_fused_strip(w->ctx, d_esa, W, rows, in->d_cj + y0, r->cond_mask, r->table_mask, b->d_arena,
                                       b->arena_cap, b->d_table, b->d_cursor, w->s_kernel) != 0)
                goto gpu_fail;
        }
        else {
            for (int k = 0; k < GCN10_N_RASTERS; k++)
                outs[k] = b->d_out[k];             /* NULL for a raster this run does not produce */
            if (g->cn_strip(w->ctx, d_esa, W, rows, in->d_cj + y0, r->cond_mask, r->table_mask, outs,
                            w->s_kernel) != 0)
                goto gpu_fail;
            /* encode the selected strips where they are */
            if (r->gpu_deflate &&
                g->deflate_strip(w->ctx, b->d_ptrs, r->n_sel, W, rows, b->d_arena, b->arena_cap,
                                 b->d_table, b->d_cursor, w->s_kernel) != 0)
                goto gpu_fail;
        }
        if (g->event_record(w->ctx, b->ev_kernel, w->s_kernel) != 0 ||
            g->stream_wait_event(w->ctx, w->s_d2h, b->ev_kernel) != 0)
            goto gpu_fail;
        /* ... and what comes back on the copy stream */
        if (r->gpu_deflate) {
            /* sizes and offsets now; the compressed bytes when drain_strip knows how many */
            int across = (W + TILE - 1) / TILE, down = (rows + TILE - 1) / TILE;

            if (g->memcpy_d2h(w->ctx, b->h_cursor, b->d_cursor, 8, w->s_d2h) != 0 ||
                g->memcpy_d2h(w->ctx, b->h_table, b->d_table,
                              (size_t)across * down * (size_t)r->n_sel * 8, w->s_d2h) != 0 ||
                g->event_record(w->ctx, b->ev_meta, w->s_d2h) != 0)
                goto gpu_fail;
        }
        else {
            for (int q = 0; q < r->n_sel; q++)
                if (g->memcpy_d2h(w->ctx, b->h_out[r->sel[q]], b->d_out[r->sel[q]], px, w->s_d2h) != 0)
                    goto gpu_fail;
            if (g->event_record(w->ctx, b->ev_d2h, w->s_d2h) != 0)
                goto gpu_fail;
        }
        b->d2h_issued = true;
        b->y0 = y0;
        b->rows = rows;

        /* While this strip is in flight, hand an earlier one to the sink: the one `drain_lag` strips back
         * (default 2).  Draining waits for that strip's kernels and counts; with a lag of 1 (rounds 1-2) the
         * worker had exactly one strip queued behind the running one and could submit the next only when the
         * previous had finished -- the trace of round 3 shows the compute stream idle ~0.5 ms between strips
         * whose kernels take 0.2-0.3 ms.  With 2 the queue holds two strips while the host waits. */
        if (s >= r->drain_lag && drain_strip(w, &w->buf[(s - r->drain_lag) % r->nbuf], tifs, W, H) != 0) {
            rc = -1;
            goto out;
        }
    }
    for (int i = 0; i < w->run->nbuf; i++)
        if (drain_strip(w, &w->buf[i], tifs, W, H) != 0) {
            rc = -1;
            goto out;
        }
    for (int i = 0; i < w->run->nbuf; i++)
        wait_sink(&w->buf[i]);
    ok = !atomic_load(&w->failed);
    if (in->n_inflate > 0) {
        /* every landcover chunk must have been a valid one (the statuses came back behind ev_ready) */
        if (g->event_sync(w->ctx, in->ev_ready) != 0)
            goto gpu_fail;
        for (size_t i = 0; i < in->n_inflate; i++)
            if (in->h_status[i] != 0) {
                wlog(w, "ERROR", true, "gdalrasterio error: cannot decode a tile of the window %d,%d %dx%d "
                                       "(stream %zu, reason %u)", in->xoff, in->yoff, W, H, i, in->h_status[i]);
                wlog(w, "ERROR", true, "esa load failed for block %d", block_id);
                ok = false;
                break;
            }
    }
    goto out;

gpu_fail:
    wlog(w, "ERROR", true, "gpu: %s", g->last_error());
    rc = -1;

out:
    /* nothing of this block may still be in flight when its buffers are reused */
    for (int i = 0; i < w->run->nbuf; i++) {
        if (w->buf[i].d2h_issued) {
            g->event_sync(w->ctx, r->gpu_deflate ? w->buf[i].ev_meta : w->buf[i].ev_d2h);
            w->buf[i].d2h_issued = false;
        }
        wait_sink(&w->buf[i]);
    }
    if (rc != 0 && w->ctx)
        g->device_sync(w->ctx);
    else if (!ok && w->ctx)
        g->stream_sync(w->ctx, w->s_kernel);    /* a block given up half-way: its kernels may still be queued */
    t_mark = now_seconds();
    {
        /* directory, close and rename of the 18 files: side by side on the I/O pool (4 ms per block in a row) */
        struct strip_buf *b0 = &w->buf[0];      /* its job counter is idle here: every strip has been drained */

        for (int k = 0; k < GCN10_N_RASTERS; k++) {
            struct finish_job *j;

            if (!tifs[k])
                continue;
            if (!ok) {
                gcn10_tiff_abort(tifs[k]);
                tifs[k] = NULL;
                continue;
            }
            j = r->pool ? malloc(sizeof *j) : NULL;
            if (!j) {
                if (gcn10_tiff_finish(tifs[k], err, sizeof err) != 0)
                    wlog(w, "ERROR", true, "%s", err);      /* "write error %d on %s", src/raster.c:221 */
                tifs[k] = NULL;
                continue;
            }
            *j = (struct finish_job){ w, b0, tifs[k] };
            tifs[k] = NULL;
            pthread_mutex_lock(&b0->mu);
            b0->pending++;
            pthread_mutex_unlock(&b0->mu);
            gcn10_pool_submit(r->pool, finish_tiff_job, j);
        }
        pthread_mutex_lock(&b0->mu);
        while (b0->pending > 0)
            pthread_cond_wait(&b0->cv, &b0->mu);
        pthread_mutex_unlock(&b0->mu);
    }
    w->t_finish += now_seconds() - t_mark;
    if (ok) {
        for (int q = 0; q < r->n_sel; q++) {
            const int k = r->sel[q];
            /* src/cn.c:366-373: one completion line and one progress line per raster */
            wlog(w, "INFO", false, "completed condition for %d: %s/%s/%s", block_id, gcn10_conds[k / 9],
                 gcn10_hcs[(k % 9) / 3], gcn10_arcs[k % 3]);
            {
                char line[256];

                snprintf(line, sizeof line, "progress: completed block %d / total %d", block_id,
                         r->n_blocks);                                  /* src/log.c:203-206 */
                gcn10_log_message(r->workers[0].log, "INFO", line, false);
            }
        }
    }
    return rc;
}

/* ------------------------------------------------------------------------ */
/* workers and the run: src/main.c:58-203                                    */
/* ------------------------------------------------------------------------ */

static void worker_teardown(struct worker *w)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    gcn10_input_teardown(w);
    if (w->ctx) {
        g->device_sync(w->ctx);
        free_strip_buffers(w);
        for (int i = 0; i < w->run->nbuf; i++) {
            struct strip_buf *b = &w->buf[i];

            if (b->ev_h2d) g->event_destroy(w->ctx, b->ev_h2d);
            if (b->ev_kernel) g->event_destroy(w->ctx, b->ev_kernel);
            if (b->ev_d2h) g->event_destroy(w->ctx, b->ev_d2h);
            if (b->ev_meta) g->event_destroy(w->ctx, b->ev_meta);
        }
        if (w->s_kernel) g->stream_destroy(w->ctx, w->s_kernel);
        if (w->s_d2h) g->stream_destroy(w->ctx, w->s_d2h);
        g->destroy(w->ctx);
        w->ctx = NULL;
    }
    gcn10_raster_close(w->esa);
    gcn10_raster_close(w->soil);
    w->esa = w->soil = NULL;
    for (int i = 0; i < w->run->nbuf; i++) {
        pthread_mutex_destroy(&w->buf[i].mu);
        pthread_cond_destroy(&w->buf[i].cv);
    }
}

/* Keeps a worker thread (and the pinned buffers it is about to allocate: first touch)
 * on the NUMA node its GPU hangs off.  Best effort: any failure leaves the thread alone. */
static void bind_to_gpu_numa_node(struct worker *w, int device)
{
    char bus[64] = "", path[256], line[4096];
    FILE *f;
    int node = -1;
    cpu_set_t set;
    char *p;

    w->device = device;
    w->numa_node = -1;
    if (w->run->gpu->pci_bus_id(device, bus, sizeof bus) != 0)
        return;
    for (char *c = bus; *c; c++)
        if (*c >= 'A' && *c <= 'F')
            *c = (char)(*c - 'A' + 'a');        /* sysfs names are lower case */
    snprintf(w->pci_bus, sizeof w->pci_bus, "%s", bus);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    f = fopen(path, "r");
    if (!f)
        return;
    if (fscanf(f, "%d", &node) != 1)
        node = -1;
    fclose(f);
    w->numa_node = node;
    if (node < 0 || getenv("GCN10_NO_NUMA_BIND"))
        return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f)
        return;
    if (!fgets(line, sizeof line, f)) {
        fclose(f);
        return;
    }
    fclose(f);
    CPU_ZERO(&set);
    for (p = line; *p;) {                       /* "0-31,64-95" */
        char *end;
        long a = strtol(p, &end, 10), b;

        if (end == p)
            break;
        b = a;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            CPU_SET((int)c, &set);
        p = (*end == ',') ? end + 1 : end;
        if (*end != ',')
            break;
    }
    if (CPU_COUNT(&set) > 0 && sched_setaffinity(0, sizeof set, &set) == 0)
        wlog(w, "INFO", false, "gpu %d (%s) is on NUMA node %d: worker bound to its %d cpus", device, bus,
             node, CPU_COUNT(&set));
}

static int worker_setup(struct worker *w)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    char err[1024];

    for (int i = 0; i < w->run->nbuf; i++) {
        pthread_mutex_init(&w->buf[i].mu, NULL);
        pthread_cond_init(&w->buf[i].cv, NULL);
        w->buf[i].owner = w;
    }
    bind_to_gpu_numa_node(w, (w->index % r->n_devices) % r->n_physical);
    if (g->init(w->device, &w->ctx) != 0) {
        wlog(w, "ERROR", true, "gpu %d: %s", w->device, g->last_error());
        return -1;
    }
    {
        char name[128];
        size_t hbm = 0;
        int cus = g->device_info(w->ctx, name, sizeof name, &hbm);

        w->n_cus = cus > 0 ? cus : 256;
    }
    /* waits for GPU events sleep between queries: the pool threads that wait for a strip's bytes, and this thread when
     * it waits for a strip's table, would otherwise spin in user mode for the whole wait (GCN10_EVENT_SLEEP_US=0: spin) */
    if (r->event_sleep_us > 0 && g->set_option)
        (void)g->set_option(w->ctx, "event_sync_sleep_us", r->event_sleep_us);
    GPU_TRY(w, g->set_tables(w->ctx, &r->tables[0][0][0], 9));
    w->fused = r->fused && g->deflate_fused_available(w->ctx) == 1;
    if (r->fused && !w->fused)
        wlog(w, "INFO", false, "the lookup tables define more than 256 pixel classes: "
                               "per-raster GPU encoding instead of the fused encoder");
    GPU_TRY(w, g->stream_create(w->ctx, &w->s_kernel));
    GPU_TRY(w, g->stream_create(w->ctx, &w->s_d2h));
    for (int i = 0; i < w->run->nbuf; i++) {
        GPU_TRY(w, g->event_create(w->ctx, &w->buf[i].ev_h2d));
        GPU_TRY(w, g->event_create(w->ctx, &w->buf[i].ev_kernel));
        GPU_TRY(w, g->event_create(w->ctx, &w->buf[i].ev_d2h));
        GPU_TRY(w, g->event_create(w->ctx, &w->buf[i].ev_meta));
    }
    /* the reference reopens both rasters for every block (src/raster.c:119);
     * here each worker keeps its own handles */
    w->esa = gcn10_raster_open(r->cfg.esa_data_path, r->cfg.esa_tile_dir, err, sizeof err);
    if (!w->esa)
        wlog(w, "ERROR", true, "%s", err);              /* "gdal open failed: ..." */
    w->soil = gcn10_raster_open(r->cfg.hysogs_data_path, NULL, err, sizeof err);
    if (!w->soil)
        wlog(w, "ERROR", true, "%s", err);
    /* the input side: its own context and stream on the same device, and (prefetch_blocks=1) its thread */
    if (gcn10_input_setup(w) != 0 || gcn10_input_start(w) != 0)
        return -1;
    return 0;
}

static void *worker_main(void *arg)
{
    struct worker *w = arg;
    struct run *r = w->run;

    if (worker_setup(w) != 0) {
        atomic_store(&r->fatal, 1);
        worker_teardown(w);
        return NULL;
    }
    for (;;) {
        /* the next block of THIS worker: staged one block ahead by its input thread, which takes the ids
         * from the run's counter (the reference's `i += size`, src/main.c:171, made dynamic) */
        struct block_in *in = gcn10_input_next(w);
        double t0;

        if (!in)
            break;
        t0 = now_seconds();
        if (in->outcome == 0 && !atomic_load(&r->fatal) && encode_block(w, in) != 0)
            atomic_store(&r->fatal, 1);     /* where the reference calls MPI_Abort */
        gcn10_abort_outputs(in);            /* files of a block that was not encoded after all (a no-op otherwise) */
        w->busy_seconds += now_seconds() - t0;
        w->blocks_done += in->outcome >= 0 ? 1 : 0;
        w->in_seq++;
        /* steady state of this worker: from the end of its first block (buffers allocated and pinned, kernels
         * loaded, workspaces grown) to the end of its last */
        if (w->in_seq == 1)
            w->t_first_done = now_seconds();
        w->t_last_done = now_seconds();
        gcn10_input_release(w, in);
        if (atomic_load(&r->fatal))
            break;
    }
    worker_teardown(w);
    w->cpu_seconds = gcn10_thread_cpu_seconds();
    return NULL;
}

/* Rank and size of this process when it was started by an MPI launcher or
 * srun, read from the launcher's environment (Open MPI, MPICH/Hydra PMI,
 * Slurm); no MPI library is linked.  Keeps `mpirun -n N gcn10 ...` meaningful
 * across nodes: one process per node, each driving its node's GPUs. */
static void outer_from_env(int *rank, int *size)
{
    static const char *const names[][2] = { { "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE" },
                                            { "PMI_RANK", "PMI_SIZE" },
                                            { "PMIX_RANK", "PMIX_SIZE" },
                                            { "SLURM_PROCID", "SLURM_NTASKS" } };

    *rank = 0;
    *size = 1;
    for (size_t i = 0; i < sizeof names / sizeof names[0]; i++) {
        const char *r = getenv(names[i][0]), *s = getenv(names[i][1]);

        if (r && s && atoi(s) > 1 && atoi(r) >= 0 && atoi(r) < atoi(s)) {
            *rank = atoi(r);
            *size = atoi(s);
            return;
        }
    }
}

/* The reference ends with MPI_Barrier and one "processed N blocks on R ranks" line from rank 0
 * (src/main.c:187-194).  The processes of an mpirun / srun launch share the log directory, not an MPI
 * library: every process leaves "<log_dir>/.done_<job>_<rank>" with its counts when its workers have
 * joined, and launcher rank 0 waits for all of them (GCN10_BARRIER_SECONDS, default 600) and logs the totals.
 * <job> tells launches apart (the launcher's job id from the environment, else the parent process id). */
static void closing_barrier(struct run *r, gcn10_log *log0)
{
    static const char *const job_vars[] = { "SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE", "PMI_JOBID",
                                            "GCN10_JOB_ID" };
    char job[96] = "", path[PATH_MAX], msg[512];
    const double limit = getenv("GCN10_BARRIER_SECONDS") ? atof(getenv("GCN10_BARRIER_SECONDS")) : 600.0;
    FILE *f;

    for (size_t i = 0; i < sizeof job_vars / sizeof job_vars[0] && !job[0]; i++)
        if (getenv(job_vars[i]) && *getenv(job_vars[i]))
            snprintf(job, sizeof job, "%s", getenv(job_vars[i]));
    if (!job[0])
        snprintf(job, sizeof job, "p%ld", (long)getppid());     /* the ranks of one mpirun share their parent */
    for (char *c = job; *c; c++)
        if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z')))
            *c = '_';
    snprintf(path, sizeof path, "%s/.done_%s_%d", r->cfg.log_dir, job, r->outer_rank);
    f = fopen(path, "w");
    if (f) {
        fprintf(f, "%d %d %d\n", r->n_blocks, r->n_workers, atomic_load(&r->fatal) ? 1 : 0);
        fclose(f);
    }
    if (r->outer_rank != 0)
        return;
    {
        int blocks = 0, ranks = 0, arrived = 0, failed = 0;
        const double t0 = now_seconds();

        for (int p = 0; p < r->outer_size; p++) {
            int b = 0, w = 0, bad = 0;
            bool got = false;

            snprintf(path, sizeof path, "%s/.done_%s_%d", r->cfg.log_dir, job, p);
            while (!got && now_seconds() - t0 < limit) {
                f = fopen(path, "r");
                if (f) {
                    got = fscanf(f, "%d %d %d", &b, &w, &bad) == 3;
                    fclose(f);
                }
                if (!got)
                    usleep(2000);
            }
            if (got) {
                arrived++;
                blocks += b;
                ranks += w;
                failed += bad;
                unlink(path);
            }
        }
        snprintf(msg, sizeof msg, "all %d processes: processed %d blocks on %d ranks%s%s", r->outer_size, blocks, ranks,
                 arrived < r->outer_size ? " (some processes did not report in time)" : "",
                 failed ? " (a process ended with an error)" : "");
        gcn10_log_message(log0, arrived < r->outer_size || failed ? "ERROR" : "INFO", msg, true);
    }
}

struct row_error_ctx {
    gcn10_log *log;
};

static void on_lookup_row_error(void *user, const char *message)
{
    struct row_error_ctx *c = user;

    gcn10_log_message(c->log, "ERROR", message, true);     /* src/cn.c:61, 71, 81 */
}

int gcn10_run(const gcn10_run_options *opt)
{
    struct run *r = calloc(1, sizeof *r);
    char err[2048] = "";
    char msg[8192];
    gcn10_log *log0 = NULL;
    int exit_code = 1, n_dev, failed_k = -1, rc;
    const char *sink = getenv("GCN10_SINK");
    double t_start = now_seconds();

    if (!r) {
        fprintf(stderr, "out of memory\n");
        return 1;
    }
    r->opt = *opt;
    if (!opt->config_path) {
        fprintf(stderr, "[rank 0] missing -c/--config <file>; see 'gcn10 -h' for usage.\n");   /* src/main.c:103-106 */
        free(r);
        return 1;
    }
    rc = gcn10_config_parse(opt->config_path, &r->cfg, err, sizeof err);
    if (rc != 0) {
        fprintf(stderr, "%s\n", err);                   /* src/config.c:52, 109-111 */
        free(r);
        return 1;
    }
    /* the rasters of this run: all 18 unless "lookups" / "conditions" (config or command line) say less */
    r->cond_mask = r->cfg.cond_mask;
    r->table_mask = r->cfg.table_mask;
    if ((opt->lookups && gcn10_parse_lookups(opt->lookups, &r->table_mask) != 0) ||
        (opt->conditions && gcn10_parse_conditions(opt->conditions, &r->cond_mask) != 0)) {
        fprintf(stderr, "[rank 0] bad --lookups / --conditions value; see 'gcn10 -h' for usage.\n");
        gcn10_config_free(&r->cfg);
        free(r);
        return 1;
    }
    for (int k = 0; k < GCN10_N_RASTERS; k++)
        if ((r->cond_mask >> (k / 9)) & 1u && (r->table_mask >> (k % 9)) & 1u)
            r->sel[r->n_sel++] = k;
    r->null_sink = sink && strcmp(sink, "null") == 0;
    /* Rows per strip.  Rounds 2 and 3 ran 768 rows (three tile rows of a 36000-px block = 423 tile positions, one round
     * of the fused statistics pass's 512 workgroup slots); larger strips lost end to end to the coarser hand-over between
     * kernels, copy-back and file writes (profiles/r02/strip_rows_ab.txt, profiles/r03/pipeline_strip_rows_48_blocks.txt).
     * With the kernels of the end of round 3 a 768-row strip is 0.11-0.22 ms of GPU work, and what a strip costs
     * besides -- a dozen HIP calls, three launch prologues, 18 writes -- weighs more: 2304 rows (nine tile rows),
     * 72 blocks, steady state, patchy blocks: null sink 0.0074 -> 0.0063 s per block, files 0.0107 -> 0.0092;
     * noisy blocks unchanged within the noise (profiles/r03/sink_sweep2_patches_72_blocks.txt,
     * strip_rows3_natural_72_blocks.txt).  3072 is no better; pinned memory grows with the strip (1.9 GB per GPU). */
    r->strip_rows = r->cfg.strip_rows > 0 ? (r->cfg.strip_rows + TILE - 1) / TILE * TILE : DEFAULT_STRIP_ROWS;
    if (r->strip_rows > MAX_STRIP_ROWS)     /* a strip's compressed-tile arena must stay below 4 GiB (32-bit offsets) */
        r->strip_rows = MAX_STRIP_ROWS;
    r->deflate_level = r->cfg.deflate_level;
    r->gpu_deflate = r->cfg.gpu_deflate != 0;
    r->fused = r->cfg.gpu_deflate == 2;
    r->gpu_inflate = r->cfg.gpu_inflate != 0;
    r->direct_io = r->cfg.direct_io != 0 || (getenv("GCN10_DIRECT_IO") && atoi(getenv("GCN10_DIRECT_IO")) != 0);
    r->prefetch = r->cfg.prefetch_blocks != 0 && !(getenv("GCN10_PREFETCH_BLOCKS") && atoi(getenv("GCN10_PREFETCH_BLOCKS")) == 0);
    r->nbuf = DEFAULT_NBUF;
    if (getenv("GCN10_STRIP_BUFFERS")) {
        int n = atoi(getenv("GCN10_STRIP_BUFFERS"));

        r->nbuf = n < 2 ? 2 : (n > MAX_NBUF ? MAX_NBUF : n);
    }
    r->drain_lag = 2;
    r->event_sleep_us = getenv("GCN10_EVENT_SLEEP_US") ? atoi(getenv("GCN10_EVENT_SLEEP_US")) : 50;
    if (getenv("GCN10_DRAIN_LAG"))
        r->drain_lag = atoi(getenv("GCN10_DRAIN_LAG"));
    if (r->drain_lag < 1)
        r->drain_lag = 1;
    if (r->drain_lag > r->nbuf - 1)
        r->drain_lag = r->nbuf - 1;     /* the strip whose buffer is reused next must have been drained */

    /* GPUs: one worker ("rank") each */
    r->gpu = gcn10_gpu_api_get(err, sizeof err);
    if (!r->gpu) {
        fprintf(stderr, "[rank 0] %s\n", err);
        goto done;
    }
    n_dev = r->gpu->device_count();
    if (n_dev <= 0) {
        fprintf(stderr, "[rank 0] no MI355X (gfx950) device visible; the CN path has no CPU fallback\n");
        goto done;
    }
    {
        /* GPUs to use, and worker threads ("ranks") per GPU: two workers per GPU keep the
         * device busy while one of them opens files, reads the soil window or finishes
         * its GeoTIFFs (measured +25 % blocks per second) */
        int gpus = opt->gpus > 0 ? opt->gpus : (r->cfg.gpus > 0 ? r->cfg.gpus : n_dev);
        int per_gpu = r->cfg.workers_per_gpu > 0 ? r->cfg.workers_per_gpu : 2;

        r->n_physical = n_dev;
        if (getenv("GCN10_REHEARSE_GPUS") && atoi(getenv("GCN10_REHEARSE_GPUS")) > 0) {
            /* dress rehearsal of an N-GPU node on the devices that are here: N logical GPUs with
             * per_gpu workers each, one I/O pool, N "timing gpu" lines; logical GPU d runs on device
             * d % n_dev.  Same code path as a real node from the block queue to the per-GPU lines. */
            r->n_devices = atoi(getenv("GCN10_REHEARSE_GPUS"));
            if (opt->gpus > 0 && opt->gpus < r->n_devices)
                r->n_devices = opt->gpus;
            r->n_workers = r->n_devices * per_gpu;
        }
        else if (getenv("GCN10_OVERSUBSCRIBE")) {    /* tests: N workers on however few GPUs there are */
            r->n_devices = n_dev;
            r->n_workers = gpus;
        }
        else {
            if (gpus > n_dev)
                gpus = n_dev;
            r->n_devices = gpus;
            r->n_workers = gpus * per_gpu;
        }
        if (r->n_workers > 64)
            r->n_workers = 64;
    }
    outer_from_env(&r->outer_rank, &r->outer_size);
    r->workers = calloc((size_t)r->n_workers, sizeof *r->workers);
    if (!r->workers)
        goto done;

    log0 = gcn10_log_open(r->cfg.log_dir, r->outer_rank * r->n_workers);   /* init_logging(rank), src/main.c:129 */
    r->workers[0].log = log0;
    snprintf(msg, sizeof msg,
             "starting processing with %d gpu workers (ranks)\n"
             "check rank_0.log in the log directory for detailed progress\n"
             "config loaded:\n"
             "  hysogs_data_path   = %s\n"
             "  esa_data_path      = %s\n"
             "  blocks_shp_path    = %s\n"
             "  lookup_table_path  = %s\n"
             "  log_dir            = %s",                /* src/main.c:114-125 */
             r->n_workers, r->cfg.hysogs_data_path, r->cfg.esa_data_path, r->cfg.blocks_shp_path,
             r->cfg.lookup_table_path, r->cfg.log_dir);
    gcn10_log_message(log0, "INFO", msg, true);
    for (int i = 1; i < r->n_workers; i++)
        r->workers[i].log = gcn10_log_open(r->cfg.log_dir, r->outer_rank * r->n_workers + i);
    for (int i = 0; i < r->n_workers; i++) {
        r->workers[i].run = r;
        r->workers[i].index = i;
        r->workers[i].rank = r->outer_rank * r->n_workers + i;
    }

    /* the block index is needed for bboxes in either mode */
    if (gcn10_blocks_open(r->cfg.blocks_shp_path, &r->blocks, err, sizeof err) != 0) {
        gcn10_log_message(log0, "ERROR", err, true);    /* "ogr open failed: ..." */
        if (!opt->blocks_file) {
            snprintf(msg, sizeof msg, "failed to read shapefile %s", r->cfg.blocks_shp_path);  /* src/main.c:143 */
            gcn10_log_message(log0, "ERROR", msg, true);
            goto done;
        }
        /* list mode: every block then fails like the reference's per-block OGROpen (src/cn.c:156-160) */
    }
    if (opt->blocks_file) {
        r->block_ids = gcn10_read_block_list(opt->blocks_file, &r->n_blocks);
        if (!r->block_ids) {
            snprintf(msg, sizeof msg, "cannot open block list file %s", opt->blocks_file);    /* src/raster.c:33 */
            gcn10_log_message(log0, "ERROR", msg, true);
        }
        if (!r->block_ids || !r->n_blocks) {
            snprintf(msg, sizeof msg, "no ids found in %s", opt->blocks_file);                /* src/main.c:135 */
            gcn10_log_message(log0, "ERROR", msg, true);
            goto done;
        }
    }
    else {
        /* every "ID" of the shapefile (what get_all_blocks is meant to return,
         * src/raster.c:94-101; its `ids[*n_blocks++]` never gets that far) */
        r->n_blocks = r->blocks.n;
        r->block_ids = malloc((size_t)(r->n_blocks ? r->n_blocks : 1) * sizeof *r->block_ids);
        if (r->block_ids)
            memcpy(r->block_ids, r->blocks.id, (size_t)r->n_blocks * sizeof *r->block_ids);
        if (!r->block_ids || !r->n_blocks) {
            snprintf(msg, sizeof msg, "no blocks found in %s", r->cfg.blocks_shp_path);       /* src/main.c:148 */
            gcn10_log_message(log0, "ERROR", msg, true);
            goto done;
        }
    }

    /* the nine lookup tables, once (the reference re-reads one per raster, src/cn.c:261) */
    {
        struct row_error_ctx ctx = { log0 };

        /* only the lookups this run produces are read (the reference reads one per raster it
         * writes, src/cn.c:261); a table that is not selected stays "no value" (255, src/cn.c:36-40) */
        rc = 0;
        for (int k = 0; k < 9 && rc == 0; k++) {
            if (!(r->table_mask & (1u << k))) {
                for (int lc = 0; lc < 256; lc++)
                    for (int sg = 0; sg < 5; sg++)
                        r->tables[k][lc][sg] = 255;
                continue;
            }
            rc = gcn10_load_lookup_table(r->cfg.lookup_table_path, gcn10_hcs[k / 3], gcn10_arcs[k % 3], r->tables[k],
                                         on_lookup_row_error, &ctx);
            if (rc != 0)
                failed_k = k;
        }
        if (rc != 0) {
            const char *hc = gcn10_hcs[failed_k / 3], *arc = gcn10_arcs[failed_k % 3];

            if (rc == -2)
                snprintf(msg, sizeof msg, "empty lookup table %s/default_lookup_%s_%s.csv",   /* src/cn.c:44 */
                         r->cfg.lookup_table_path, hc, arc);
            else if (rc == -3)
                snprintf(msg, sizeof msg, "lookup table path too long: %s/default_lookup_%s_%s.csv", /* :23 */
                         r->cfg.lookup_table_path, hc, arc);
            else
                snprintf(msg, sizeof msg, "cannot open lookup table %s/default_lookup_%s_%s.csv",    /* :30 */
                         r->cfg.lookup_table_path, hc, arc);
            gcn10_log_message(log0, "ERROR", msg, true);
            goto done;
        }
    }

    snprintf(msg, sizeof msg, "processing %d blocks %s", r->n_blocks,
             opt->blocks_file ? "from list file" : "from shapefile");      /* src/main.c:165-167 */
    gcn10_log_message(log0, "INFO", msg, true);
    if (r->n_workers > r->n_blocks) {       /* no worker (and no pinned memory) without a block */
        for (int i = r->n_blocks > 0 ? r->n_blocks : 1; i < r->n_workers; i++) {
            gcn10_log_close(r->workers[i].log);
            r->workers[i].log = NULL;
        }
        r->n_workers = r->n_blocks > 0 ? r->n_blocks : 1;
    }

    /* under mpirun / srun every process takes the reference's static share of the
     * list, i = rank, rank + size, ... (src/main.c:171), and feeds its own GPUs */
    if (r->outer_size > 1) {
        int kept = 0;

        for (int i = r->outer_rank; i < r->n_blocks; i += r->outer_size)
            r->block_ids[kept++] = r->block_ids[i];
        snprintf(msg, sizeof msg, "process %d of %d: %d of %d blocks", r->outer_rank, r->outer_size,
                 kept, r->n_blocks);
        gcn10_log_message(log0, "INFO", msg, true);
        r->n_blocks = kept;
    }

    {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        int nthreads = r->cfg.io_threads > 0 ? r->cfg.io_threads : (int)(ncpu > 2 ? ncpu - 1 : 1);

        if (nthreads > 64)
            nthreads = 64;
        r->pool = gcn10_pool_create(nthreads);
        if (!r->pool) {
            gcn10_log_message(log0, "ERROR", "cannot start the tile compression threads", true);
            goto done;
        }
    }

    atomic_store(&r->next_block, 0);
    atomic_store(&r->fatal, 0);
    for (int i = 0; i < r->n_workers; i++)
        if (pthread_create(&r->workers[i].thread, NULL, worker_main, &r->workers[i]) != 0) {
            gcn10_log_message(log0, "ERROR", "cannot start a gpu worker thread", true);
            atomic_store(&r->fatal, 1);
            r->n_workers = i;
            break;
        }
    for (int i = 0; i < r->n_workers; i++)
        pthread_join(r->workers[i].thread, NULL);      /* MPI_Barrier, src/main.c:187 */

    snprintf(msg, sizeof msg, "processed %d blocks on %d ranks", r->n_blocks, r->n_workers);  /* src/main.c:191 */
    gcn10_log_message(log0, "INFO", msg, true);
    if (r->outer_size > 1)
        closing_barrier(r, log0);           /* MPI_Barrier + rank 0's summary, src/main.c:187-194 */
    {
        int done_blocks = 0;
        double busy = 0, rd = 0, gw = 0, sw = 0, so = 0, cr = 0, fi = 0, dv = 0, steady = 0, iw = 0, ib = 0;

        for (int i = 0; i < r->n_workers; i++) {
            done_blocks += r->workers[i].blocks_done;
            busy += r->workers[i].busy_seconds;
            rd += r->workers[i].t_read;
            gw += r->workers[i].t_gpu_wait;
            sw += r->workers[i].t_sink_wait;
            so += r->workers[i].t_soil;
            cr += r->workers[i].t_create;
            fi += r->workers[i].t_finish;
            dv += r->workers[i].t_device;
            iw += r->workers[i].t_in_wait;
            ib += r->workers[i].t_in_busy;
        }
        {
            /* from the moment the first worker had its GPU, buffers and files ready: what a long
             * run sees per block */
            double first = 0.0;

            for (int i = 0; i < r->n_workers; i++)
                if (r->workers[i].t_first_block > 0.0 && (first == 0.0 || r->workers[i].t_first_block < first))
                    first = r->workers[i].t_first_block;
            steady = first > 0.0 ? now_seconds() - first : 0.0;
        }
        {
            /* blocks per second of every worker after its first block, added up: the rate a long run sees */
            double rate = 0.0;

            for (int i = 0; i < r->n_workers; i++) {
                const struct worker *w = &r->workers[i];

                if (w->in_seq > 1 && w->t_last_done > w->t_first_done)
                    rate += (double)(w->in_seq - 1) / (w->t_last_done - w->t_first_done);
            }
            if (rate > 0.0) {
                snprintf(msg, sizeof msg, "timing: steady state %.4f s per block (%.2f blocks per second: every worker's blocks "
                         "after its first, which pays for buffers, pinning and workspaces)", 1.0 / rate, rate);
                gcn10_log_message(log0, "INFO", msg, false);
            }
        }
        snprintf(msg, sizeof msg, "timing: %d blocks, %.3f s wall (%.3f s after start-up), %d gpu worker(s)%s%s%s; worker seconds: "
                 "in blocks %.3f, reading landcover %.3f, waiting for gpu %.3f, waiting for sink %.3f, "
                 "soil window %.3f, creating outputs %.3f, finishing outputs %.3f, device setup %.3f, "
                 "waiting for input %.3f; input thread seconds: %.3f%s",
                 done_blocks, now_seconds() - t_start, steady, r->n_workers, r->null_sink ? ", null sink" : "",
                 r->gpu_deflate ? (r->fused ? ", fused gpu deflate" : ", gpu deflate") : ", host zlib",
                 r->gpu_inflate ? ", gpu inflate of deflate landcover" : "", busy, rd, gw, sw, so, cr, fi, dv, iw, ib,
                 r->prefetch ? " (one block ahead of the encoder)" : " (in turn with the encoder)");
        gcn10_log_message(log0, "INFO", msg, false);
        {
            /* CPU seconds of the whole process (all threads): against the wall time this says whether the host
             * side is what bounds the run (a cgroup CPU quota shows as user + system ~ quota x wall) */
            struct rusage ru;

            if (getrusage(RUSAGE_SELF, &ru) == 0) {
                const double cpu = (double)ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 +
                                   (double)ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;

                snprintf(msg, sizeof msg, "timing: host cpu seconds: user %.3f, system %.3f, over %.3f s wall (%.3f per block); "
                         "voluntary / involuntary context switches %ld / %ld; pinned host memory allocated %.1f MB; "
                         "peak resident set %.1f MB",
                         (double)ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6,
                         (double)ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6, now_seconds() - t_start,
                         done_blocks > 0 ? cpu / done_blocks : 0.0, ru.ru_nvcsw, ru.ru_nivcsw,
                         (double)atomic_load(&r->pinned_bytes) / 1e6, (double)ru.ru_maxrss / 1e3);
                gcn10_log_message(log0, "INFO", msg, false);
                {
                    /* ... by kind of thread; the rest is the HIP runtime's own threads and the main thread */
                    double enc = 0.0, inp = 0.0;
                    const double io = gcn10_pool_cpu_seconds(r->pool);

                    for (int i = 0; i < r->n_workers; i++) {
                        enc += r->workers[i].cpu_seconds;
                        inp += r->workers[i].in_cpu_seconds;
                    }
                    snprintf(msg, sizeof msg, "timing: host cpu seconds by thread: encoder workers %.3f, input threads %.3f, "
                             "i/o pool %.3f, others (HIP runtime, main) %.3f", enc, inp, io, cpu - enc - inp - io);
                    gcn10_log_message(log0, "INFO", msg, false);
                    {
                        long n_put = 0, n_gate = 0, n_fin = 0, n_row = 0;
                        const double c_put = gcn10_pool_cpu_of(r->pool, put_tiles_job, &n_put);
                        const double c_gate = gcn10_pool_cpu_of(r->pool, strip_gate_job, &n_gate);
                        const double c_fin = gcn10_pool_cpu_of(r->pool, finish_tiff_job, &n_fin);
                        const double c_row = gcn10_pool_cpu_of(r->pool, tile_row_job, &n_row);

                        snprintf(msg, sizeof msg, "timing: i/o pool cpu seconds by job: appending a raster's strip %.3f (%ld), "
                                 "waiting for a strip's bytes %.3f (%ld), finishing files %.3f (%ld), host deflate %.3f (%ld), "
                                 "reading and decoding inputs %.3f", c_put, n_put, c_gate, n_gate, c_fin, n_fin, c_row, n_row,
                                 io - c_put - c_gate - c_fin - c_row);
                        gcn10_log_message(log0, "INFO", msg, false);
                        {
                            uint64_t hits = 0, misses = 0;
                            size_t held = 0;

                            gcn10_tiff_cache_stats(&hits, &misses, &held);
                            snprintf(msg, sizeof msg, "timing: decoded-chunk cache (soil strips): %llu hits, %llu misses, %.1f MB held",
                                     (unsigned long long)hits, (unsigned long long)misses, (double)held / 1e6);
                            gcn10_log_message(log0, "INFO", msg, false);
                        }
                    }
                }
            }
        }
        /* one line per GPU: when a node's GPUs are not equally busy, these tell a slow device or PCIe
         * root complex (waiting for gpu) from a slow NUMA node or file system (reading, waiting for sink) */
        for (int d = 0; d < r->n_devices; d++) {
            int nw = 0, nb = 0;
            double b_ = 0, rd_ = 0, gw_ = 0, sw_ = 0, so_ = 0, fi_ = 0;
            const struct worker *first = NULL;

            for (int i = 0; i < r->n_workers; i++) {
                const struct worker *w = &r->workers[i];

                if (w->index % r->n_devices != d)
                    continue;
                if (!first)
                    first = w;
                nw++;
                nb += w->blocks_done;
                b_ += w->busy_seconds;
                rd_ += w->t_read;
                gw_ += w->t_gpu_wait;
                sw_ += w->t_sink_wait;
                so_ += w->t_soil;
                fi_ += w->t_finish;
            }
            if (!first)
                continue;
            snprintf(msg, sizeof msg, "timing gpu %d (pci %s, numa node %d): %d worker(s), %d blocks, worker seconds: "
                     "in blocks %.3f, reading landcover %.3f, waiting for gpu %.3f, waiting for sink %.3f, "
                     "soil window %.3f, finishing outputs %.3f",
                     d, first->pci_bus[0] ? first->pci_bus : "?", first->numa_node, nw, nb, b_, rd_, gw_, sw_, so_, fi_);
            gcn10_log_message(log0, "INFO", msg, false);
        }
    }
    exit_code = atomic_load(&r->fatal) ? 1 : 0;

done:
    gcn10_pool_destroy(r->pool);
    if (r->workers) {
        for (int i = 1; i < r->n_workers; i++)
            gcn10_log_close(r->workers[i].log);
    }
    gcn10_log_close(log0);                              /* finalize_logging, src/main.c:197 */
    free(r->workers);
    free(r->block_ids);
    gcn10_blocks_free(&r->blocks);
    gcn10_config_free(&r->cfg);
    free(r);
    return exit_code;
}

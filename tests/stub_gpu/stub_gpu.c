/* stub_gpu.c -- a host-only stand-in for libgcn10_gpu.so.  TEST INFRASTRUCTURE ONLY.
 *
 * Purpose: run the threaded host program (gcn10_amd/csrc/host/pipeline.c: block workers, strip-buffer
 * hand-over, sink pool) under ThreadSanitizer, which needs no GPU (`make tsan-host`).  It implements
 * the entry points of include/gcn10_gpu.h that the pipeline binds (gpuapi.c) with plain loops, and --
 * so that the sanitizer sees the same ordering problem the real device poses -- with ASYNCHRONOUS
 * streams: every stream is a thread that executes its operations in order, events are recorded and
 * waited for as on the device, and a "device" copy reads or writes the pinned host buffer at the
 * time the stream reaches it, not when the call returns.  A buffer reused before its event has
 * fired is then a data race the sanitizer reports.
 *
 * Loaded only through GCN10_GPU_LIB in tests / make tsan-host.  It is never installed, never found
 * by the default search of gpuapi.c, and the product does not fall back to it: without a GPU
 * library the product stops (tests/test_abi.py).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "gcn10_gpu.h"

#include <pthread.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ---- streams and events --------------------------------------------------------------------- */
struct stub_event {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    unsigned long long recorded, completed;    /* instances recorded / reached by their stream */
};

struct op {
    struct op *next;
    void (*fn)(void *);
    void *arg;
};

struct stub_stream {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    struct op *head, *tail;
    unsigned long long queued, done;
    bool quit;
    struct stub_stream *next_in_ctx;
};

struct gcn10_gpu_ctx {
    pthread_mutex_t mu;
    struct stub_stream *streams;
    struct stub_stream *main_stream;
    uint8_t T8[9][6][256];
    int n_tables;
    /* prepared tile */
    uint8_t *codes;                 /* [hsy][W]: drained plane | undrained plane << 4 */
    int tile_W, tile_hsy;
};

static __thread char g_err[256];

static void *stream_main(void *arg)
{
    struct stub_stream *s = arg;

    for (;;) {
        struct op *o;

        pthread_mutex_lock(&s->mu);
        while (!s->head && !s->quit)
            pthread_cond_wait(&s->cv, &s->mu);
        if (!s->head && s->quit) {
            pthread_mutex_unlock(&s->mu);
            return NULL;
        }
        o = s->head;
        s->head = o->next;
        if (!s->head)
            s->tail = NULL;
        pthread_mutex_unlock(&s->mu);
        o->fn(o->arg);
        free(o);
        pthread_mutex_lock(&s->mu);
        s->done++;
        pthread_cond_broadcast(&s->cv);
        pthread_mutex_unlock(&s->mu);
    }
}

static struct stub_stream *stream_new(gcn10_gpu_ctx *ctx)
{
    struct stub_stream *s = calloc(1, sizeof *s);

    if (!s)
        return NULL;
    pthread_mutex_init(&s->mu, NULL);
    pthread_cond_init(&s->cv, NULL);
    if (pthread_create(&s->th, NULL, stream_main, s) != 0) {
        free(s);
        return NULL;
    }
    pthread_mutex_lock(&ctx->mu);
    s->next_in_ctx = ctx->streams;
    ctx->streams = s;
    pthread_mutex_unlock(&ctx->mu);
    return s;
}

static int enqueue(struct stub_stream *s, void (*fn)(void *), void *arg)
{
    struct op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    o->next = NULL;
    o->fn = fn;
    o->arg = arg;
    pthread_mutex_lock(&s->mu);
    if (s->tail)
        s->tail->next = o;
    else
        s->head = o;
    s->tail = o;
    s->queued++;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
    return GCN10_OK;
}

static void stream_drain(struct stub_stream *s)
{
    pthread_mutex_lock(&s->mu);
    const unsigned long long want = s->queued;
    while (s->done < want)
        pthread_cond_wait(&s->cv, &s->mu);
    pthread_mutex_unlock(&s->mu);
}

static struct stub_stream *as_stream(gcn10_gpu_ctx *ctx, gcn10_stream_t st)
{
    return st ? (struct stub_stream *)st : ctx->main_stream;
}

/* ---- C ABI ---------------------------------------------------------------------------------- */
int gcn10_gpu_abi_version(void) { return GCN10_GPU_ABI_VERSION; }
const char *gcn10_gpu_last_error(void) { return g_err; }
int gcn10_gpu_device_count(void) { return 1; }

int gcn10_gpu_init(int device, gcn10_gpu_ctx **out)
{
    gcn10_gpu_ctx *ctx;

    if (device != 0 || !out) {
        snprintf(g_err, sizeof g_err, "stub: device %d", device);
        return GCN10_E_INVAL;
    }
    ctx = calloc(1, sizeof *ctx);
    if (!ctx)
        return GCN10_E_NOMEM;
    pthread_mutex_init(&ctx->mu, NULL);
    ctx->main_stream = stream_new(ctx);
    if (!ctx->main_stream) {
        free(ctx);
        return GCN10_E_NOMEM;
    }
    *out = ctx;
    return GCN10_OK;
}

int gcn10_gpu_device_sync(gcn10_gpu_ctx *ctx)
{
    pthread_mutex_lock(&ctx->mu);
    struct stub_stream *s = ctx->streams;
    pthread_mutex_unlock(&ctx->mu);
    for (; s; s = s->next_in_ctx)
        stream_drain(s);
    return GCN10_OK;
}

void gcn10_gpu_destroy(gcn10_gpu_ctx *ctx)
{
    if (!ctx)
        return;
    gcn10_gpu_device_sync(ctx);
    for (struct stub_stream *s = ctx->streams; s;) {
        struct stub_stream *n = s->next_in_ctx;

        pthread_mutex_lock(&s->mu);
        s->quit = true;
        pthread_cond_broadcast(&s->cv);
        pthread_mutex_unlock(&s->mu);
        pthread_join(s->th, NULL);
        free(s);
        s = n;
    }
    free(ctx->codes);
    free(ctx);
}

int gcn10_gpu_device_info(gcn10_gpu_ctx *ctx, char *name, size_t cap, size_t *hbm)
{
    (void)ctx;
    if (name && cap)
        snprintf(name, cap, "host stub (tests only)");
    if (hbm)
        *hbm = 0;
    return 1;
}

int gcn10_gpu_pci_bus_id(int device, char *buf, size_t cap)
{
    (void)device;
    snprintf(buf, cap, "0000:00:00.0");
    return GCN10_OK;
}

int gcn10_gpu_malloc(gcn10_gpu_ctx *c, size_t n, void **p) { (void)c; *p = malloc(n ? n : 1); return *p ? GCN10_OK : GCN10_E_NOMEM; }
int gcn10_gpu_soil_words_state(gcn10_gpu_ctx *c, gcn10_stream_t s) { (void)c; (void)s; return 0; }
int gcn10_gpu_free(gcn10_gpu_ctx *c, void *p) { (void)c; free(p); return GCN10_OK; }
int gcn10_gpu_host_alloc(gcn10_gpu_ctx *c, size_t n, void **p) { (void)c; *p = malloc(n ? n : 1); return *p ? GCN10_OK : GCN10_E_NOMEM; }
int gcn10_gpu_host_free(gcn10_gpu_ctx *c, void *p) { (void)c; free(p); return GCN10_OK; }

struct copy_op { void *dst; const void *src; size_t n; int value; };
static void do_copy(void *a) { struct copy_op *c = a; memcpy(c->dst, c->src, c->n); free(c); }
static void do_set(void *a) { struct copy_op *c = a; memset(c->dst, c->value, c->n); free(c); }

static int queue_copy(gcn10_gpu_ctx *ctx, void *dst, const void *src, size_t n, int value, bool set, gcn10_stream_t st)
{
    struct copy_op *c = malloc(sizeof *c);

    if (!c)
        return GCN10_E_NOMEM;
    *c = (struct copy_op){ dst, src, n, value };
    return enqueue(as_stream(ctx, st), set ? do_set : do_copy, c);
}
int gcn10_gpu_memcpy_h2d(gcn10_gpu_ctx *c, void *d, const void *s, size_t n, gcn10_stream_t st) { return n ? queue_copy(c, d, s, n, 0, false, st) : GCN10_OK; }
int gcn10_gpu_memcpy_d2h(gcn10_gpu_ctx *c, void *d, const void *s, size_t n, gcn10_stream_t st) { return n ? queue_copy(c, d, s, n, 0, false, st) : GCN10_OK; }
int gcn10_gpu_memset(gcn10_gpu_ctx *c, void *d, int v, size_t n, gcn10_stream_t st) { return n ? queue_copy(c, d, NULL, n, v, true, st) : GCN10_OK; }

int gcn10_gpu_stream_create(gcn10_gpu_ctx *ctx, gcn10_stream_t *st)
{
    *st = stream_new(ctx);
    return *st ? GCN10_OK : GCN10_E_NOMEM;
}
int gcn10_gpu_stream_destroy(gcn10_gpu_ctx *ctx, gcn10_stream_t st) { (void)ctx; stream_drain(st); return GCN10_OK; }   /* freed with the context */
int gcn10_gpu_stream_sync(gcn10_gpu_ctx *ctx, gcn10_stream_t st) { stream_drain(as_stream(ctx, st)); return GCN10_OK; }

int gcn10_gpu_event_create(gcn10_gpu_ctx *ctx, gcn10_event_t *ev)
{
    struct stub_event *e = calloc(1, sizeof *e);

    (void)ctx;
    if (!e)
        return GCN10_E_NOMEM;
    pthread_mutex_init(&e->mu, NULL);
    pthread_cond_init(&e->cv, NULL);
    *ev = e;
    return GCN10_OK;
}
int gcn10_gpu_event_destroy(gcn10_gpu_ctx *ctx, gcn10_event_t ev) { (void)ctx; free(ev); return GCN10_OK; }

struct ev_op { struct stub_event *e; unsigned long long instance; };
static void do_signal(void *a)
{
    struct ev_op *o = a;

    pthread_mutex_lock(&o->e->mu);
    if (o->e->completed < o->instance)
        o->e->completed = o->instance;
    pthread_cond_broadcast(&o->e->cv);
    pthread_mutex_unlock(&o->e->mu);
    free(o);
}
static void do_wait(void *a)
{
    struct ev_op *o = a;

    pthread_mutex_lock(&o->e->mu);
    while (o->e->completed < o->instance)
        pthread_cond_wait(&o->e->cv, &o->e->mu);
    pthread_mutex_unlock(&o->e->mu);
    free(o);
}
int gcn10_gpu_event_record(gcn10_gpu_ctx *ctx, gcn10_event_t ev, gcn10_stream_t st)
{
    struct stub_event *e = ev;
    struct ev_op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    pthread_mutex_lock(&e->mu);
    o->e = e;
    o->instance = ++e->recorded;
    pthread_mutex_unlock(&e->mu);
    return enqueue(as_stream(ctx, st), do_signal, o);
}
int gcn10_gpu_event_sync(gcn10_gpu_ctx *ctx, gcn10_event_t ev)
{
    struct stub_event *e = ev;

    (void)ctx;
    pthread_mutex_lock(&e->mu);
    const unsigned long long want = e->recorded;
    while (e->completed < want)
        pthread_cond_wait(&e->cv, &e->mu);
    pthread_mutex_unlock(&e->mu);
    return GCN10_OK;
}
int gcn10_gpu_stream_wait_event(gcn10_gpu_ctx *ctx, gcn10_stream_t st, gcn10_event_t ev)
{
    struct stub_event *e = ev;
    struct ev_op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    pthread_mutex_lock(&e->mu);
    o->e = e;
    o->instance = e->recorded;      /* the instance recorded so far, as hipStreamWaitEvent */
    pthread_mutex_unlock(&e->mu);
    return enqueue(as_stream(ctx, st), do_wait, o);
}
int gcn10_gpu_event_elapsed_ms(gcn10_gpu_ctx *c, gcn10_event_t a, gcn10_event_t b, float *ms) { (void)c; (void)a; (void)b; *ms = 0.f; return GCN10_OK; }

/* ---- the CN path, plain loops (src/cn.c:88-131, 218-232 as DESIGN.md section 3 splits them) ---- */
int gcn10_gpu_set_tables(gcn10_gpu_ctx *ctx, const int *tables, int n)
{
    memset(ctx->T8, 255, sizeof ctx->T8);
    for (int k = 0; k < n; k++)
        for (int lc = 0; lc < 256; lc++)
            for (int s = 0; s < 5; s++) {
                int v = tables[(k * 256 + lc) * 5 + s];
                ctx->T8[k][s][lc] = v < 255 ? (uint8_t)v : 255;
            }
    ctx->n_tables = n;
    return GCN10_OK;
}

struct prep_op { gcn10_gpu_ctx *ctx; const uint8_t *coarse; int hsx, hsy; const int32_t *ci; int W; };
static void do_prepare(void *a)
{
    struct prep_op *o = a;
    gcn10_gpu_ctx *ctx = o->ctx;

    free(ctx->codes);
    ctx->codes = malloc((size_t)o->hsy * o->W + 1);
    ctx->tile_W = o->W;
    ctx->tile_hsy = o->hsy;
    for (int r = 0; r < o->hsy; r++)
        for (int x = 0; x < o->W; x++) {
            uint8_t h = o->coarse[(size_t)r * o->hsx + o->ci[x]];
            int dual = h >= 11 && h <= 14;
            int d = dual ? 4 : (h < 5 ? h : 5), u = dual ? h - 10 : (h < 5 ? h : 5);
            ctx->codes[(size_t)r * o->W + x] = (uint8_t)(d | (u << 4));
        }
    free(o);
}
int gcn10_gpu_prepare_tile(gcn10_gpu_ctx *ctx, const uint8_t *coarse, int hsx, int hsy, const int32_t *ci, int W,
                           gcn10_stream_t st)
{
    struct prep_op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    *o = (struct prep_op){ ctx, coarse, hsx, hsy, ci, W };
    return enqueue(as_stream(ctx, st), do_prepare, o);
}

struct strip_op { gcn10_gpu_ctx *ctx; const uint8_t *esa; int W, rows; const int32_t *cj; unsigned cm, tm; uint8_t *out[GCN10_N_RASTERS]; };
static void do_strip(void *a)
{
    struct strip_op *o = a;
    gcn10_gpu_ctx *ctx = o->ctx;

    for (int y = 0; y < o->rows; y++) {
        const uint8_t *codes = ctx->codes + (size_t)o->cj[y] * o->W;
        for (int x = 0; x < o->W; x++) {
            const size_t i = (size_t)y * o->W + x;
            for (int c = 0; c < 2; c++) {
                if (!(o->cm & (1u << c)))
                    continue;
                const int s = (codes[x] >> (4 * c)) & 15;
                for (int k = 0; k < 9; k++)
                    if (o->tm & (1u << k))
                        o->out[c * 9 + k][i] = ctx->T8[k][s][o->esa[i]];
            }
        }
    }
    free(o);
}
int gcn10_gpu_cn_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows, const int32_t *cj, unsigned cm,
                       unsigned tm, uint8_t *const out[GCN10_N_RASTERS], gcn10_stream_t st)
{
    struct strip_op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    *o = (struct strip_op){ ctx, esa, W, rows, cj, cm, tm, { 0 } };
    memcpy(o->out, out, sizeof o->out);
    return enqueue(as_stream(ctx, st), do_strip, o);
}

/* ---- tile encode / decode with stock zlib --------------------------------------------------- */
size_t gcn10_gpu_deflate_arena_bound(int W, int rows, int n_rasters)
{
    size_t tiles = (size_t)((W + 255) / 256) * (size_t)((rows + 255) / 256);
    return tiles * (size_t)n_rasters * (compressBound(65536) + 64);
}

struct enc_op { const uint8_t *const *rasters; int n, W, rows; uint8_t *arena; size_t cap; uint32_t *table; unsigned long long *cursor; };
static void do_encode(void *a)
{
    struct enc_op *o = a;
    const int across = (o->W + 255) / 256, down = (o->rows + 255) / 256;
    static __thread uint8_t tile[65536];
    size_t used = 0;

    for (int r = 0; r < o->n; r++)
        for (int ty = 0; ty < down; ty++)
            for (int tx = 0; tx < across; tx++) {
                uLongf len;
                uint32_t *ent = o->table + ((size_t)r * across * down + (size_t)ty * across + tx) * 2;

                memset(tile, 0, sizeof tile);
                for (int y = 0; y < 256 && ty * 256 + y < o->rows; y++) {
                    int w = o->W - tx * 256 < 256 ? o->W - tx * 256 : 256;
                    memcpy(tile + y * 256, o->rasters[r] + (size_t)(ty * 256 + y) * o->W + tx * 256, (size_t)w);
                }
                len = (uLongf)(o->cap - used);
                if (used >= o->cap || compress2(o->arena + used, &len, tile, sizeof tile, 1) != Z_OK) {
                    ent[0] = 0xffffffffu;
                    ent[1] = 0;
                    continue;
                }
                ent[0] = (uint32_t)used;
                ent[1] = (uint32_t)len;
                used += (len + 15) & ~(size_t)15;
            }
    *o->cursor = used;
    free(o);
}
int gcn10_gpu_deflate_strip(gcn10_gpu_ctx *ctx, const uint8_t *const *rasters, int n, int W, int rows, uint8_t *arena,
                            size_t cap, uint32_t *table, unsigned long long *cursor, gcn10_stream_t st)
{
    struct enc_op *o = malloc(sizeof *o);

    if (!o)
        return GCN10_E_NOMEM;
    *o = (struct enc_op){ rasters, n, W, rows, arena, cap, table, cursor };
    return enqueue(as_stream(ctx, st), do_encode, o);
}

int gcn10_gpu_deflate_fused_available(gcn10_gpu_ctx *ctx) { (void)ctx; return 0; }     /* the per-raster path is the one stubbed */
int gcn10_gpu_deflate_fused_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows, const int32_t *cj, unsigned cm,
                                  unsigned tm, uint8_t *arena, size_t cap, uint32_t *table, unsigned long long *cursor,
                                  gcn10_stream_t st)
{
    (void)ctx; (void)esa; (void)W; (void)rows; (void)cj; (void)cm; (void)tm; (void)arena; (void)cap; (void)table; (void)cursor; (void)st;
    snprintf(g_err, sizeof g_err, "stub: no fused encoder");
    return GCN10_E_STATE;
}

struct inf_op { const uint8_t *comp; const gcn10_inflate_tile *tiles; int n; uint8_t *dst; size_t stride; uint32_t *status; };
static void do_inflate(void *a)
{
    struct inf_op *o = a;

    for (int i = 0; i < o->n; i++) {
        const gcn10_inflate_tile *t = &o->tiles[i];
        uint8_t *tmp = calloc(1, (size_t)(t->out_len ? t->out_len : 1) + t->chunk_w);
        uLongf len = t->out_len;
        int rc = Z_MEM_ERROR;

        if (tmp && (t->flags & GCN10_TILE_RAW)) {
            rc = t->in_len >= t->out_len ? Z_OK : Z_DATA_ERROR;
            if (rc == Z_OK)
                memcpy(tmp, o->comp + t->in_off, t->out_len);
        }
        else if (tmp) {
            rc = uncompress(tmp, &len, o->comp + t->in_off, t->in_len);
        }
        o->status[i] = (rc == Z_OK || rc == Z_BUF_ERROR) ? 0 : GCN10_INFLATE_E_CODE;
        if (tmp && (t->flags & GCN10_TILE_PREDICTOR2) && t->chunk_w)
            for (size_t r0 = 0; r0 < t->out_len; r0 += t->chunk_w)          /* per chunk row, from its first byte */
                for (size_t x = 1; x < t->chunk_w && r0 + x < t->out_len; x++)
                    tmp[r0 + x] = (uint8_t)(tmp[r0 + x] + tmp[r0 + x - 1]);
        for (uint32_t y = 0; tmp && y < t->copy_h; y++)
            memcpy(o->dst + t->dst_off + (size_t)y * o->stride, tmp + (size_t)(t->src_y + y) * t->chunk_w + t->src_x,
                   t->copy_w);
        free(tmp);
    }
    free(o);
}
int gcn10_gpu_inflate_tiles(gcn10_gpu_ctx *ctx, const uint8_t *comp, const gcn10_inflate_tile *tiles, int n,
                            uint32_t chunk_bytes, uint8_t *dst, size_t stride, uint32_t *status, gcn10_stream_t st)
{
    struct inf_op *o = malloc(sizeof *o);

    (void)chunk_bytes;
    if (!o)
        return GCN10_E_NOMEM;
    *o = (struct inf_op){ comp, tiles, n, dst, stride, status };
    return enqueue(as_stream(ctx, st), do_inflate, o);
}

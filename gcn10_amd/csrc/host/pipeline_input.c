/* pipeline_input.c -- the input side of a block worker: one block ahead of the encoder.
 *
 * What it replaces: the front half of process_block() (/root/reference/src/cn.c:155-232): the
 * block's bbox, load_raster() of the landcover window (src/raster.c:106-189, one blocking
 * GDALRasterIO into a malloc'ed buffer) and of the soil window, and the index arithmetic of the
 * resample.  In the reference that is serial with everything else a rank does.  Here every block
 * worker has an input thread with a GPU context and stream of its own: while the worker encodes
 * block N, the thread
 *   - takes the next block id from the run's counter and computes its windows (geo.c, bit exact),
 *   - reads the soil window and builds the index maps into pinned memory, copies them to HBM,
 *   - PLANS the landcover window (raster.c / tiff.c: which chunks of which files, clipped how),
 *     has the I/O pool pread the chunks -- compressed or raw, as they lie in the files -- into
 *     a ring of pinned buffers, and copies batch k to HBM (hipMemcpyAsync on its stream) while the
 *     pool fills batch k+1: the "pinned-host staging, double-buffered" of the north star,
 *   - launches gcn10_gpu_inflate_tiles on them: DEFLATE tiles are decoded, raw tiles untiled, TIFF
 *     predictor 2 undone, all in HBM, into the row-major window d_block,
 *   - (windows the GPU side cannot take -- LZW, PackBits, overlapping mosaic sources -- are read by
 *     the host reader strip by strip into the same ring and copied up the same way),
 *   - records one event behind all of it and hands the slot to the worker.
 * Two slots per worker: the block being encoded and the one being staged, so at most two blocks'
 * input are resident per worker.  prefetch_blocks=0 runs the same code on the worker thread itself.
 */
#include "pipeline_internal.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define now_seconds gcn10_now_seconds
#define wlog gcn10_wlog

enum { RING_BYTES = 64 << 20 };     /* one pinned staging buffer; a block of DEFLATE landcover is 2-3 of them */

/* chunks of a read plan -> pinned staging, a slice per pool job */
struct comp_job {
    const struct gcn10_chunk_ref *chunks;
    const gcn10_inflate_tile *jobs;
    size_t n;
    uint8_t *dst;                   /* staging buffer; chunk i goes to dst + (jobs[i].in_off - base) */
    uint64_t base;
    pthread_mutex_t *mu;
    pthread_cond_t *cv;
    int *pending, *failed;
};

static void comp_job_run(void *arg)
{
    struct comp_job *j = arg;
    int bad = 0;

    for (size_t i = 0; i < j->n && !bad; i++) {
        uint8_t *p = j->dst + (j->jobs[i].in_off - j->base);
        size_t left = j->chunks[i].nbytes;
        uint64_t off = j->chunks[i].file_off;

        while (left > 0) {
            ssize_t got = pread(j->chunks[i].fd, p, left, (off_t)off);

            if (got <= 0) {
                bad = 1;
                break;
            }
            p += got;
            off += (uint64_t)got;
            left -= (size_t)got;
        }
        memset(p, 0, 16);               /* the decoder's bit reader may look a few bytes ahead */
    }
    pthread_mutex_lock(j->mu);
    if (bad)
        *j->failed = 1;
    if (--*j->pending == 0)
        pthread_cond_broadcast(j->cv);
    pthread_mutex_unlock(j->mu);
    free(j);
}

/* longest stream first among the compressed ones (one workgroup decodes one stream and the GPU hands
 * workgroups out in index order: the stragglers start first); raw chunks keep their file order behind them */
static int by_size_desc(const void *a, const void *b)
{
    const struct gcn10_chunk_ref *x = a, *y = b;
    const int xr = (x->flags & GCN10_TILE_RAW) != 0, yr = (y->flags & GCN10_TILE_RAW) != 0;

    if (xr != yr)
        return xr - yr;
    if (xr)
        return x->file_off < y->file_off ? -1 : (x->file_off > y->file_off ? 1 : 0);
    return x->nbytes < y->nbytes ? 1 : (x->nbytes > y->nbytes ? -1 : 0);
}

#define GPU_IN(w, call)                                                        \
    do {                                                                       \
        if ((call) != 0) {                                                     \
            wlog((w), "ERROR", true, "gpu: %s", (w)->run->gpu->last_error());  \
            return -2;                                                         \
        }                                                                      \
    } while (0)

/* a staging buffer of the ring, free to be written: its last copy to the device has finished */
static int ring_take(struct worker *w, int k)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    if (w->ring_busy[k]) {
        GPU_IN(w, g->event_sync(w->in_ctx, w->ev_ring[k]));
        w->ring_busy[k] = false;
    }
    return 0;
}

static int ring_sent(struct worker *w, int k)
{
    GPU_IN(w, w->run->gpu->event_record(w->in_ctx, w->ev_ring[k], w->s_in));
    w->ring_busy[k] = true;
    return 0;
}

static int ring_ensure(struct worker *w, size_t need)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    if (need <= w->ring_cap)
        return 0;
    for (int k = 0; k < N_RING; k++) {
        if (ring_take(w, k) != 0)
            return -2;
        if (w->h_ring[k])
            g->host_free(w->in_ctx, w->h_ring[k]);
        w->h_ring[k] = NULL;
    }
    w->ring_cap = 0;
    need = (need + 4095) & ~(size_t)4095;
    for (int k = 0; k < N_RING; k++)
    {
        GPU_IN(w, g->host_alloc(w->in_ctx, need, (void **)&w->h_ring[k]));
        atomic_fetch_add(&w->run->pinned_bytes, (long long)need);
    }
    w->ring_cap = need;
    return 0;
}

/* The landcover window through the GPU side.  0 = issued on s_in; 1 = this window needs the host reader;
 * -1 = the window cannot be read (logged; the block is skipped as after a failed load_raster,
 * src/cn.c:188-192); -2 = device error (logged; fatal for the run). */
static int stage_planned(struct worker *w, struct block_in *in)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    struct gcn10_read_plan plan;
    char err[1024] = "";
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    int rc, k = 0;
    size_t comp_bytes = 0, largest = 0;
    const int W = in->W, H = in->H;
    double t0 = now_seconds();

    rc = gcn10_raster_plan_window(w->esa, in->xoff, in->yoff, W, H, &plan, err, sizeof err);
    if (rc > 0)
        return 1;
    if (rc < 0) {
        wlog(w, "ERROR", true, "%s", err);
        wlog(w, "ERROR", true, "esa load failed for block %d", in->block_id);
        return -1;
    }
    if (!getenv("GCN10_INFLATE_FILE_ORDER"))      /* (A/B switch for tools/) */
        qsort(plan.chunks, plan.n, sizeof *plan.chunks, by_size_desc);
    rc = -2;
    if (plan.n > in->jobs_cap) {
        if (in->h_jobs) g->host_free(w->in_ctx, in->h_jobs);
        if (in->d_jobs) g->free(w->in_ctx, in->d_jobs);
        if (in->h_status) g->host_free(w->in_ctx, in->h_status);
        if (in->d_status) g->free(w->in_ctx, in->d_status);
        in->h_jobs = NULL;
        in->d_jobs = NULL;
        in->h_status = NULL;
        in->d_status = NULL;
        in->jobs_cap = 0;
        if (g->host_alloc(w->in_ctx, plan.n * sizeof *in->h_jobs, (void **)&in->h_jobs) != 0 ||
            g->malloc(w->in_ctx, plan.n * sizeof *in->d_jobs, (void **)&in->d_jobs) != 0 ||
            g->host_alloc(w->in_ctx, plan.n * 4, (void **)&in->h_status) != 0 ||
            g->malloc(w->in_ctx, plan.n * 4, (void **)&in->d_status) != 0)
            goto gpu_fail;
        in->jobs_cap = plan.n;
    }
    for (size_t i = 0; i < plan.n; i++) {
        const struct gcn10_chunk_ref *c = &plan.chunks[i];
        gcn10_inflate_tile *j = &in->h_jobs[i];
        const size_t slot = (((size_t)c->nbytes + 15) & ~(size_t)15) + 16;

        j->in_off = comp_bytes;
        j->in_len = c->nbytes;
        j->out_len = c->out_len;
        j->chunk_w = c->chunk_w;
        j->src_x = c->src_x;
        j->src_y = c->src_y;
        j->copy_w = c->copy_w;
        j->copy_h = c->copy_h;
        j->flags = c->flags;
        j->dst_off = (uint64_t)c->dst_y * (uint64_t)W + c->dst_x;
        comp_bytes += slot;
        if (slot > largest)
            largest = slot;
        in->h_status[i] = 0xffffffffu;
    }
    if (gcn10_ensure_dev_on(w, w->in_ctx, (void **)&in->d_comp, &in->d_comp_cap, comp_bytes + 16) != 0 ||
        ring_ensure(w, largest > (size_t)RING_BYTES ? largest : (size_t)RING_BYTES) != 0)
        goto out;               /* logged */
    if (plan.covered < (uint64_t)W * (uint64_t)H &&
        g->memset(w->in_ctx, in->d_block, 0, (size_t)W * (size_t)H, w->s_in) != 0)
        goto gpu_fail;

    /* batches of chunks: the pool preads batch b into staging buffer b % N_RING while the copy of batch
     * b - 1 to the device is in flight */
    for (size_t i0 = 0; i0 < plan.n;) {
        const uint64_t base = in->h_jobs[i0].in_off;
        size_t i1 = i0;
        uint64_t end = base;
        int pending = 0, failed = 0;

        while (i1 < plan.n) {
            const uint64_t e = in->h_jobs[i1].in_off + ((((uint64_t)in->h_jobs[i1].in_len + 15) & ~(uint64_t)15) + 16);

            if (e - base > w->ring_cap)
                break;
            end = e;
            i1++;
        }
        if (ring_take(w, k) != 0)
            goto out;
        for (size_t i = i0; i < i1; i += 32) {
            struct comp_job *j = malloc(sizeof *j);
            struct comp_job job = { plan.chunks + i, in->h_jobs + i, i1 - i < 32 ? i1 - i : 32, w->h_ring[k], base,
                                    &mu, &cv, &pending, &failed };

            pthread_mutex_lock(&mu);
            pending++;
            pthread_mutex_unlock(&mu);
            if (!j || !r->pool) {
                struct comp_job *tmp = j ? j : malloc(sizeof *tmp);

                if (!tmp) {
                    pthread_mutex_lock(&mu);
                    pending--;
                    failed = 1;
                    pthread_mutex_unlock(&mu);
                    break;
                }
                *tmp = job;
                comp_job_run(tmp);
                continue;
            }
            *j = job;
            gcn10_pool_submit(r->pool, comp_job_run, j);
        }
        pthread_mutex_lock(&mu);
        while (pending > 0)
            pthread_cond_wait(&cv, &mu);
        pthread_mutex_unlock(&mu);
        if (failed) {
            w->t_read += now_seconds() - t0;
            wlog(w, "ERROR", true, "gdalrasterio error: cannot read the landcover tiles of the window %d,%d %dx%d",
                 in->xoff, in->yoff, W, H);
            wlog(w, "ERROR", true, "esa load failed for block %d", in->block_id);
            rc = -1;
            goto out;
        }
        if (g->memcpy_h2d(w->in_ctx, in->d_comp + base, w->h_ring[k], (size_t)(end - base), w->s_in) != 0)
            goto gpu_fail;
        if (ring_sent(w, k) != 0)
            goto out;
        k = (k + 1) % N_RING;
        i0 = i1;
    }
    w->t_read += now_seconds() - t0;
    if (plan.n > 0 &&
        (g->memcpy_h2d(w->in_ctx, in->d_jobs, in->h_jobs, plan.n * sizeof *in->h_jobs, w->s_in) != 0 ||
         g->memcpy_h2d(w->in_ctx, in->d_status, in->h_status, plan.n * 4, w->s_in) != 0 ||
         g->inflate_tiles(w->in_ctx, in->d_comp, in->d_jobs, (int)plan.n, plan.max_chunk_bytes ? plan.max_chunk_bytes : 16,
                          in->d_block, (size_t)W, in->d_status, w->s_in) != 0 ||
         g->memcpy_d2h(w->in_ctx, in->h_status, in->d_status, plan.n * 4, w->s_in) != 0))
        goto gpu_fail;
    in->n_inflate = plan.n;
    rc = 0;
    goto out;

gpu_fail:
    /* a device error (allocation, copy, launch) is not a failed load_raster: it ends the run like every
     * other GPU error of process_block, instead of skipping the block */
    wlog(w, "ERROR", true, "gpu: %s", g->last_error());
    rc = -2;
out:
    /* the files may close: the bytes are in pinned memory or further */
    gcn10_read_plan_free(&plan);
    return rc;
}

/* The landcover window through the host reader, a band of rows at a time: decoded by the I/O pool straight
 * into a pinned staging buffer (replaces the malloc + GDALRasterIO of src/raster.c:169-178), copied up
 * while the next band is decoded.  0, -1 (logged, block skipped) or -2 (device error). */
static int stage_host(struct worker *w, struct block_in *in)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    const int W = in->W, H = in->H;
    char err[1024] = "";
    int band, k = 0;

    if (ring_ensure(w, (size_t)W * 256 > (size_t)RING_BYTES ? (size_t)W * 256 : (size_t)RING_BYTES) != 0)
        return -2;
    band = (int)(w->ring_cap / (size_t)W);
    if (band > r->strip_rows)
        band = r->strip_rows;           /* the granularity of round 1-2's per-strip reads */
    for (int y0 = 0; y0 < H; y0 += band) {
        const int rows = H - y0 < band ? H - y0 : band;
        double t0 = now_seconds();
        int rc;

        if (ring_take(w, k) != 0)
            return -2;
        rc = gcn10_raster_read_mt(w->esa, in->xoff, in->yoff + y0, W, rows, w->h_ring[k], r->pool, err, sizeof err);
        w->t_read += now_seconds() - t0;
        if (rc != 0) {
            wlog(w, "ERROR", true, "%s", err);
            wlog(w, "ERROR", true, "esa load failed for block %d", in->block_id);
            return -1;
        }
        GPU_IN(w, g->memcpy_h2d(w->in_ctx, in->d_block + (size_t)y0 * (size_t)W, w->h_ring[k],
                                (size_t)W * (size_t)rows, w->s_in));
        if (ring_sent(w, k) != 0)
            return -2;
        k = (k + 1) % N_RING;
    }
    return 0;
}

/* One block from its id to "everything it needs is on its way to HBM": the front half of process_block
 * (src/cn.c:155-232).  Returns the slot's outcome: 0, 1 (skipped, logged) or -1 (fatal). */
static int fill_block(struct worker *w, struct block_in *in, int block_id)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    char err[1024] = "";
    double esa_t[6], soil_t[6], soil_gt[6];
    int esa_rx, esa_ry, soil_rx, soil_ry, sxoff, syoff, bi, rc;
    double t_mark;

    in->block_id = block_id;
    in->n_inflate = 0;
    if (!w->esa) {
        wlog(w, "ERROR", true, "esa load failed for block %d", block_id);          /* src/cn.c:189 */
        return 1;
    }
    if (!w->soil) {
        wlog(w, "ERROR", true, "hysogs load failed for block %d", block_id);
        return 1;
    }
    /* block geometry: attribute filter "ID"=<id>, first feature (src/cn.c:162-184) */
    bi = gcn10_blocks_find(&r->blocks, block_id);
    if (bi < 0) {
        wlog(w, "ERROR", true, "block %d not found", block_id);                     /* src/cn.c:173 */
        return 1;
    }
    /* landcover window (src/cn.c:187-192, src/raster.c:126-162) */
    gcn10_raster_info(w->esa, &esa_rx, &esa_ry, esa_t);
    if (gcn10_raster_window(esa_t, esa_rx, esa_ry, r->blocks.bbox[bi], &in->xoff, &in->yoff, &in->W, &in->H,
                            in->gt) != 0) {
        wlog(w, "ERROR", true, "invalid raster bounds for %s", r->cfg.esa_data_path);   /* src/raster.c:143 */
        wlog(w, "ERROR", true, "esa load failed for block %d", block_id);               /* src/cn.c:189 */
        return 1;
    }
    /* soil window (src/cn.c:195-203) */
    gcn10_raster_info(w->soil, &soil_rx, &soil_ry, soil_t);
    if (gcn10_raster_window(soil_t, soil_rx, soil_ry, r->blocks.bbox[bi], &sxoff, &syoff, &in->hsx, &in->hsy,
                            soil_gt) != 0) {
        wlog(w, "ERROR", true, "invalid raster bounds for %s", r->cfg.hysogs_data_path);
        wlog(w, "ERROR", true, "hysogs load failed for block %d", block_id);            /* src/cn.c:198 */
        return 1;
    }
    if ((long long)in->W * (long long)in->H > INT_MAX) {
        /* the reference's int npix (src/cn.c:208) overflows here; refuse instead */
        wlog(w, "ERROR", true, "block %d: %d x %d pixels exceed 2^31-1", block_id, in->W, in->H);
        return 1;
    }
    /* soil window and index maps in pinned memory owned by the slot: their copies to the GPU need no wait */
    if (gcn10_ensure_pinned_on(w, w->in_ctx, (void **)&in->h_coarse, &in->h_coarse_cap, (size_t)in->hsx * (size_t)in->hsy) != 0 ||
        gcn10_ensure_pinned_on(w, w->in_ctx, (void **)&in->h_ci, &in->h_ci_cap, (size_t)in->W * sizeof *in->h_ci) != 0 ||
        gcn10_ensure_pinned_on(w, w->in_ctx, (void **)&in->h_cj, &in->h_cj_cap, (size_t)in->H * sizeof *in->h_cj) != 0) {
        wlog(w, "ERROR", true, "malloc failed for hysogs resampling, block %d", block_id); /* src/cn.c:211 */
        return -1;
    }
    t_mark = now_seconds();
    /* the soil raster is a global file of full-width strips: a block's window touches ~1440 of
     * them, decoded concurrently on the I/O pool and only as far as the window reaches */
    if (gcn10_raster_read_mt(w->soil, sxoff, syoff, in->hsx, in->hsy, in->h_coarse, r->pool, err, sizeof err) != 0) {
        wlog(w, "ERROR", true, "%s", err);
        wlog(w, "ERROR", true, "hysogs load failed for block %d", block_id);
        return 1;
    }
    gcn10_build_index_maps(in->gt, soil_gt, in->W, in->H, in->hsx, in->hsy, in->h_ci, in->h_cj);   /* src/cn.c:218-229 */
    w->t_soil += now_seconds() - t_mark;

    if (gcn10_ensure_dev_on(w, w->in_ctx, (void **)&in->d_coarse, &in->coarse_cap, (size_t)in->hsx * (size_t)in->hsy) != 0 ||
        gcn10_ensure_dev_on(w, w->in_ctx, (void **)&in->d_ci, &in->ci_cap, (size_t)in->W * 4) != 0 ||
        gcn10_ensure_dev_on(w, w->in_ctx, (void **)&in->d_cj, &in->cj_cap, (size_t)in->H * 4) != 0 ||
        gcn10_ensure_dev_on(w, w->in_ctx, (void **)&in->d_block, &in->block_cap, (size_t)in->W * (size_t)in->H) != 0)
        return -1;
    if (g->memcpy_h2d(w->in_ctx, in->d_coarse, in->h_coarse, (size_t)in->hsx * (size_t)in->hsy, w->s_in) != 0 ||
        g->memcpy_h2d(w->in_ctx, in->d_ci, in->h_ci, (size_t)in->W * 4, w->s_in) != 0 ||
        g->memcpy_h2d(w->in_ctx, in->d_cj, in->h_cj, (size_t)in->H * 4, w->s_in) != 0) {
        wlog(w, "ERROR", true, "gpu: %s", g->last_error());
        return -1;
    }
    /* the landcover window: chunks as they lie in the files -> HBM -> decoded / untiled there; what the GPU
     * side cannot take, and everything with gpu_inflate=0, through the host reader */
    rc = r->gpu_inflate ? stage_planned(w, in) : 1;
    if (rc == 1)
        rc = stage_host(w, in);
    if (rc == 0 && g->event_record(w->in_ctx, in->ev_ready, w->s_in) != 0) {
        wlog(w, "ERROR", true, "gpu: %s", g->last_error());
        rc = -2;
    }
    if (rc != 0) {
        /* nothing of a block that is not going to be encoded may still be in flight when its slot is reused */
        g->stream_sync(w->in_ctx, w->s_in);
        for (int k = 0; k < N_RING; k++)
            w->ring_busy[k] = false;
        return rc == -2 ? -1 : 1;
    }
    /* the block's 18 files, while the copies and the decoder run (the reference creates each raster's file
     * inside save_raster, after computing it: src/raster.c:204; an existing raster of that name is untouched
     * until this one is complete either way: files are written as <name>.part) */
    rc = gcn10_create_outputs(w, in);
    if (rc < 0)
        return -1;
    return 0;               /* (rc == 1: logged; the worker finds tifs_ok false and gives the block up) */
}

/* takes the next block of the queue and fills slot `in` with it; false at the end of the queue */
static bool take_and_fill(struct worker *w, struct block_in *in)
{
    struct run *r = w->run;
    int i = atomic_fetch_add(&r->next_block, 1);
    double t0;

    if (i >= r->n_blocks || atomic_load(&r->fatal))
        return false;
    wlog(w, "INFO", true, "processing block %d", r->block_ids[i]);          /* src/main.c:172-173 */
    t0 = now_seconds();
    if (w->t_first_block == 0.0)
        w->t_first_block = t0;
    in->outcome = fill_block(w, in, r->block_ids[i]);
    w->t_in_busy += now_seconds() - t0;
    if (in->outcome < 0)
        atomic_store(&r->fatal, 1);     /* where the reference calls MPI_Abort */
    return true;
}

static void *input_main(void *arg)
{
    struct worker *w = arg;

    for (int n = 0;; n++) {
        struct block_in *in = &w->in[n % N_IN];
        bool more;

        pthread_mutex_lock(&w->in_mu);
        while (in->state != IN_FREE && !w->in_stop)
            pthread_cond_wait(&w->in_cv, &w->in_mu);
        if (w->in_stop) {
            pthread_mutex_unlock(&w->in_mu);
            { w->in_cpu_seconds = gcn10_thread_cpu_seconds(); return NULL; }
        }
        in->state = IN_FILLING;
        pthread_mutex_unlock(&w->in_mu);

        more = take_and_fill(w, in);

        pthread_mutex_lock(&w->in_mu);
        in->state = more ? IN_READY : IN_END;
        pthread_cond_broadcast(&w->in_cv);
        pthread_mutex_unlock(&w->in_mu);
        if (!more)
            { w->in_cpu_seconds = gcn10_thread_cpu_seconds(); return NULL; }
    }
}

int gcn10_input_setup(struct worker *w)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    pthread_mutex_init(&w->in_mu, NULL);
    pthread_cond_init(&w->in_cv, NULL);
    /* a context of its own: the input side calls the GPU library concurrently with the worker, and a
     * context is not shared between threads (include/gcn10_gpu.h); device memory and events are valid
     * in both, they belong to the device */
    GPU_IN(w, g->init(w->device, &w->in_ctx));
    if (w->run->event_sleep_us > 0 && g->set_option)
        (void)g->set_option(w->in_ctx, "event_sync_sleep_us", w->run->event_sleep_us);
    GPU_IN(w, g->stream_create(w->in_ctx, &w->s_in));
    for (int k = 0; k < N_RING; k++)
        GPU_IN(w, g->event_create(w->in_ctx, &w->ev_ring[k]));
    for (int k = 0; k < N_IN; k++)
        GPU_IN(w, g->event_create(w->in_ctx, &w->in[k].ev_ready));
    return 0;
}

int gcn10_input_start(struct worker *w)
{
    if (!w->run->prefetch)
        return 0;
    if (pthread_create(&w->in_thread, NULL, input_main, w) != 0) {
        wlog(w, "ERROR", true, "cannot start the input thread");
        return -1;
    }
    w->in_thread_started = true;
    return 0;
}

struct block_in *gcn10_input_next(struct worker *w)
{
    struct block_in *in;

    if (!w->in_thread_started) {
        /* no input thread: the worker stages its own block, in turn with encoding it */
        in = &w->in[0];
        if (!take_and_fill(w, in))
            return NULL;
        in->state = IN_READY;
        return in;
    }
    in = &w->in[w->in_seq % N_IN];
    {
        double t0 = now_seconds();

        pthread_mutex_lock(&w->in_mu);
        while (in->state != IN_READY && in->state != IN_END)
            pthread_cond_wait(&w->in_cv, &w->in_mu);
        pthread_mutex_unlock(&w->in_mu);
        w->t_in_wait += now_seconds() - t0;
    }
    return in->state == IN_READY ? in : NULL;
}

void gcn10_input_release(struct worker *w, struct block_in *in)
{
    pthread_mutex_lock(&w->in_mu);
    in->state = IN_FREE;
    pthread_cond_broadcast(&w->in_cv);
    pthread_mutex_unlock(&w->in_mu);
}

void gcn10_input_stop(struct worker *w)
{
    if (!w->in_thread_started)
        return;
    pthread_mutex_lock(&w->in_mu);
    w->in_stop = true;
    pthread_cond_broadcast(&w->in_cv);
    pthread_mutex_unlock(&w->in_mu);
    pthread_join(w->in_thread, NULL);
    w->in_thread_started = false;
}

void gcn10_input_teardown(struct worker *w)
{
    const struct gcn10_gpu_api *g = w->run->gpu;

    gcn10_input_stop(w);
    if (w->in_ctx) {
        g->device_sync(w->in_ctx);
        for (int k = 0; k < N_IN; k++) {
            struct block_in *in = &w->in[k];

            gcn10_abort_outputs(in);        /* a staged block the worker never took */
            if (in->h_coarse) g->host_free(w->in_ctx, in->h_coarse);
            if (in->h_ci) g->host_free(w->in_ctx, in->h_ci);
            if (in->h_cj) g->host_free(w->in_ctx, in->h_cj);
            if (in->d_coarse) g->free(w->in_ctx, in->d_coarse);
            if (in->d_ci) g->free(w->in_ctx, in->d_ci);
            if (in->d_cj) g->free(w->in_ctx, in->d_cj);
            if (in->d_block) g->free(w->in_ctx, in->d_block);
            if (in->d_comp) g->free(w->in_ctx, in->d_comp);
            if (in->h_jobs) g->host_free(w->in_ctx, in->h_jobs);
            if (in->d_jobs) g->free(w->in_ctx, in->d_jobs);
            if (in->h_status) g->host_free(w->in_ctx, in->h_status);
            if (in->d_status) g->free(w->in_ctx, in->d_status);
            if (in->ev_ready) g->event_destroy(w->in_ctx, in->ev_ready);
            memset(in, 0, sizeof *in);
        }
        for (int k = 0; k < N_RING; k++) {
            if (w->h_ring[k]) g->host_free(w->in_ctx, w->h_ring[k]);
            if (w->ev_ring[k]) g->event_destroy(w->in_ctx, w->ev_ring[k]);
            w->h_ring[k] = NULL;
            w->ev_ring[k] = NULL;
        }
        if (w->s_in) g->stream_destroy(w->in_ctx, w->s_in);
        g->destroy(w->in_ctx);
        w->in_ctx = NULL;
    }
    pthread_mutex_destroy(&w->in_mu);
    pthread_cond_destroy(&w->in_cv);
}

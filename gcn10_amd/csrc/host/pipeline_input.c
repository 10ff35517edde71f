/* pipeline_input.c -- landcover input of a block through the GPU decoder.
 *
 * What it replaces: load_raster()'s GDALRasterIO (/root/reference/src/raster.c:167-176) reads
 * and inflates the block's landcover window on the host.  Here the window is *planned*
 * (raster.c / tiff.c: which compressed tiles or strips, from which files, clipped how), the
 * compressed chunks are read into pinned memory by the I/O pool, one H2D carries them, and
 * gcn10_gpu_inflate_tiles decodes every stream in HBM and places the window row-major.
 */
#include "pipeline_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define now_seconds gcn10_now_seconds
#define wlog gcn10_wlog
#define ensure_dev gcn10_ensure_dev

/* compressed chunks of a read plan -> pinned staging, a slice per pool job */
struct comp_job {
    const struct gcn10_chunk_ref *chunks;
    const gcn10_inflate_tile *jobs;
    size_t n;
    uint8_t *dst;
    pthread_mutex_t *mu;
    pthread_cond_t *cv;
    int *pending, *failed;
};

static void comp_job_run(void *arg)
{
    struct comp_job *j = arg;
    int bad = 0;

    for (size_t i = 0; i < j->n && !bad; i++) {
        uint8_t *p = j->dst + j->jobs[i].in_off;
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

/* longest compressed stream first */
static int by_size_desc(const void *a, const void *b)
{
    const struct gcn10_chunk_ref *x = a, *y = b;

    return x->nbytes < y->nbytes ? 1 : (x->nbytes > y->nbytes ? -1 : 0);
}

int gcn10_inflate_block(struct worker *w, int xoff, int yoff, int W, int H, int block_id)
{
    struct run *r = w->run;
    const struct gcn10_gpu_api *g = r->gpu;
    struct gcn10_read_plan plan;
    char err[1024] = "";
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    int pending = 0, failed = 0, rc;
    size_t comp_bytes = 0;
    double t0 = now_seconds();

    rc = gcn10_raster_plan_window(w->esa, xoff, yoff, W, H, &plan, err, sizeof err);
    if (rc > 0)
        return 1;
    if (rc < 0) {
        wlog(w, "ERROR", true, "%s", err);
        wlog(w, "ERROR", true, "esa load failed for block %d", block_id);
        return -1;
    }
    /* one workgroup decodes one stream and the GPU hands workgroups out in index order as slots free
     * up: with the longest streams first the last, partly filled round of slots (a block's ~1300 tiles
     * over 1024 slots) decodes the short ones instead of waiting for one long straggler */
    if (!getenv("GCN10_INFLATE_FILE_ORDER"))      /* (A/B switch for tools/) */
        qsort(plan.chunks, plan.n, sizeof *plan.chunks, by_size_desc);
    rc = -1;
    if (plan.n > w->jobs_cap) {
        if (w->h_jobs) g->host_free(w->ctx, w->h_jobs);
        if (w->d_jobs) g->free(w->ctx, w->d_jobs);
        if (w->h_status) g->host_free(w->ctx, w->h_status);
        if (w->d_status) g->free(w->ctx, w->d_status);
        w->h_jobs = NULL;
        w->d_jobs = NULL;
        w->h_status = NULL;
        w->d_status = NULL;
        w->jobs_cap = 0;
        if (g->host_alloc(w->ctx, plan.n * sizeof *w->h_jobs, (void **)&w->h_jobs) != 0 ||
            g->malloc(w->ctx, plan.n * sizeof *w->d_jobs, (void **)&w->d_jobs) != 0 ||
            g->host_alloc(w->ctx, plan.n * 4, (void **)&w->h_status) != 0 ||
            g->malloc(w->ctx, plan.n * 4, (void **)&w->d_status) != 0)
            goto gpu_fail;
        w->jobs_cap = plan.n;
    }
    for (size_t i = 0; i < plan.n; i++) {
        const struct gcn10_chunk_ref *c = &plan.chunks[i];
        gcn10_inflate_tile *j = &w->h_jobs[i];

        j->in_off = comp_bytes;
        j->in_len = c->nbytes;
        j->out_len = c->chunk_w * c->rows;
        j->chunk_w = c->chunk_w;
        j->src_x = c->src_x;
        j->src_y = c->src_y;
        j->copy_w = c->copy_w;
        j->copy_h = c->copy_h;
        j->reserved = 0;
        j->dst_off = (uint64_t)c->dst_y * (uint64_t)W + c->dst_x;
        comp_bytes += (((size_t)c->nbytes + 15) & ~(size_t)15) + 16;
        w->h_status[i] = 0xffffffffu;
    }
    if (comp_bytes > w->h_comp_cap) {
        size_t cap = comp_bytes + comp_bytes / 4 + 4096;

        if (w->h_comp) g->host_free(w->ctx, w->h_comp);
        if (w->d_comp) g->free(w->ctx, w->d_comp);
        w->h_comp = NULL;
        w->d_comp = NULL;
        w->h_comp_cap = w->d_comp_cap = 0;
        if (g->host_alloc(w->ctx, cap, (void **)&w->h_comp) != 0 ||
            g->malloc(w->ctx, cap, (void **)&w->d_comp) != 0)
            goto gpu_fail;
        w->h_comp_cap = w->d_comp_cap = cap;
    }
    if (ensure_dev(w, (void **)&w->d_block, &w->block_cap, (size_t)W * (size_t)H) != 0) {
        rc = -2;        /* ensure_dev logged it: a device allocation failed */
        goto out;
    }
    /* compressed bytes: a few dozen chunks per pool job */
    for (size_t i = 0; i < plan.n; i += 32) {
        struct comp_job *j = malloc(sizeof *j);
        struct comp_job job = { plan.chunks + i, w->h_jobs + i, plan.n - i < 32 ? plan.n - i : 32, w->h_comp,
                                &mu, &cv, &pending, &failed };

        if (!j || !r->pool) {
            struct comp_job *tmp = j ? j : malloc(sizeof *tmp);

            if (!tmp) {
                failed = 1;
                break;
            }
            *tmp = job;
            pthread_mutex_lock(&mu);
            pending++;
            pthread_mutex_unlock(&mu);
            comp_job_run(tmp);
            continue;
        }
        *j = job;
        pthread_mutex_lock(&mu);
        pending++;
        pthread_mutex_unlock(&mu);
        gcn10_pool_submit(r->pool, comp_job_run, j);
    }
    pthread_mutex_lock(&mu);
    while (pending > 0)
        pthread_cond_wait(&cv, &mu);
    pthread_mutex_unlock(&mu);
    w->t_read += now_seconds() - t0;
    if (failed) {
        wlog(w, "ERROR", true, "gdalrasterio error: cannot read the landcover tiles of the window %d,%d %dx%d",
             xoff, yoff, W, H);
        wlog(w, "ERROR", true, "esa load failed for block %d", block_id);
        goto out;
    }
    if (plan.n == 0) {
        if (g->memset(w->ctx, w->d_block, 0, (size_t)W * (size_t)H, w->s_kernel) != 0 ||
            g->event_record(w->ctx, w->ev_inflate, w->s_kernel) != 0)
            goto gpu_fail;
        rc = 0;
        goto out;
    }
    if (g->memcpy_h2d(w->ctx, w->d_comp, w->h_comp, comp_bytes, w->s_h2d) != 0 ||
        g->memcpy_h2d(w->ctx, w->d_jobs, w->h_jobs, plan.n * sizeof *w->h_jobs, w->s_h2d) != 0 ||
        g->memcpy_h2d(w->ctx, w->d_status, w->h_status, plan.n * 4, w->s_h2d) != 0 ||
        g->event_record(w->ctx, w->ev_comp, w->s_h2d) != 0 ||
        g->stream_wait_event(w->ctx, w->s_kernel, w->ev_comp) != 0 ||
        (plan.covered < (uint64_t)W * (uint64_t)H &&
         g->memset(w->ctx, w->d_block, 0, (size_t)W * (size_t)H, w->s_kernel) != 0) ||
        g->inflate_tiles(w->ctx, w->d_comp, w->d_jobs, (int)plan.n, plan.max_chunk_bytes, w->d_block,
                         (size_t)W, w->d_status, w->s_kernel) != 0 ||
        g->memcpy_d2h(w->ctx, w->h_status, w->d_status, plan.n * 4, w->s_kernel) != 0 ||
        g->event_record(w->ctx, w->ev_inflate, w->s_kernel) != 0)
        goto gpu_fail;
    w->n_inflate = plan.n;
    rc = 0;
    goto out;

gpu_fail:
    /* a device error (allocation, copy, launch) is not a failed load_raster: the caller ends the
     * run with it, like every other GPU error of process_block, instead of skipping the block */
    wlog(w, "ERROR", true, "gpu: %s", g->last_error());
    rc = -2;
out:
    /* the files may close: the compressed bytes are in pinned memory now */
    gcn10_read_plan_free(&plan);
    return rc;
}


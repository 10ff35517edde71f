/* pool.c -- a fixed pool of threads that compress raster tiles.
 *
 * GDAL deflates every 256x256 block inside the one blocking GDALRasterIO call of
 * save_raster() (/root/reference/src/raster.c:217-219), on the rank's only
 * thread; the paper names this as what dominates the run
 * (paper/paper.md:152-153).  Here the tiles of all 18 rasters of a strip are
 * independent jobs for all host cores while the GPU works on the next strip.
 */
#include "host_internal.h"

#include <pthread.h>
#include <time.h>
#include <stdlib.h>

struct job {
    gcn10_job_fn fn;
    void *arg;
    struct job *next;
};

struct gcn10_pool {
    pthread_t *threads;
    int n;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    struct job *head, *tail;
    bool stop;
    struct { gcn10_job_fn fn; double cpu; long n; } by_fn[16];   /* CPU time by kind of job (timing lines) */
};

double gcn10_thread_cpu_seconds(void);

static void *pool_main(void *arg)
{
    gcn10_pool *p = arg;

    for (;;) {
        struct job *j;

        pthread_mutex_lock(&p->mu);
        while (!p->head && !p->stop)
            pthread_cond_wait(&p->cv, &p->mu);
        if (!p->head) {             /* stop requested and queue drained */
            pthread_mutex_unlock(&p->mu);
            return NULL;
        }
        j = p->head;
        p->head = j->next;
        if (!p->head)
            p->tail = NULL;
        pthread_mutex_unlock(&p->mu);
        {
            const double c0 = gcn10_thread_cpu_seconds();
            const gcn10_job_fn fn = j->fn;

            j->fn(j->arg);
            free(j);
            {
                const double used = gcn10_thread_cpu_seconds() - c0;

                pthread_mutex_lock(&p->mu);
                for (int k = 0; k < 16; k++)
                    if (p->by_fn[k].fn == fn || !p->by_fn[k].fn) {
                        p->by_fn[k].fn = fn;
                        p->by_fn[k].cpu += used;
                        p->by_fn[k].n++;
                        break;
                    }
                pthread_mutex_unlock(&p->mu);
            }
        }
    }
}

double gcn10_thread_cpu_seconds(void)
{
    struct timespec ts;

    return clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + ts.tv_nsec * 1e-9 : 0.0;
}

double gcn10_pool_cpu_of(gcn10_pool *p, gcn10_job_fn fn, long *n_jobs)
{
    double cpu = 0.0;

    if (n_jobs)
        *n_jobs = 0;
    if (!p)
        return 0.0;
    pthread_mutex_lock(&p->mu);
    for (int k = 0; k < 16; k++)
        if (p->by_fn[k].fn == fn) {
            cpu = p->by_fn[k].cpu;
            if (n_jobs)
                *n_jobs = p->by_fn[k].n;
        }
    pthread_mutex_unlock(&p->mu);
    return cpu;
}

double gcn10_pool_cpu_seconds(gcn10_pool *p)
{
    double sum = 0.0;

    for (int i = 0; p && i < p->n; i++) {
        clockid_t cid;
        struct timespec ts;

        if (pthread_getcpuclockid(p->threads[i], &cid) == 0 && clock_gettime(cid, &ts) == 0)
            sum += (double)ts.tv_sec + ts.tv_nsec * 1e-9;
    }
    return sum;
}

gcn10_pool *gcn10_pool_create(int n_threads)
{
    gcn10_pool *p = calloc(1, sizeof *p);

    if (!p)
        return NULL;
    if (n_threads < 1)
        n_threads = 1;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv, NULL);
    p->threads = calloc((size_t)n_threads, sizeof *p->threads);
    if (!p->threads) {
        free(p);
        return NULL;
    }
    for (int i = 0; i < n_threads; i++) {
        if (pthread_create(&p->threads[i], NULL, pool_main, p) != 0)
            break;
        p->n++;
    }
    if (p->n == 0) {
        gcn10_pool_destroy(p);
        return NULL;
    }
    return p;
}

void gcn10_pool_submit(gcn10_pool *p, gcn10_job_fn fn, void *arg)
{
    struct job *j = malloc(sizeof *j);

    if (!j) {               /* no memory for the queue node: run it here */
        fn(arg);
        return;
    }
    j->fn = fn;
    j->arg = arg;
    j->next = NULL;
    pthread_mutex_lock(&p->mu);
    if (p->tail)
        p->tail->next = j;
    else
        p->head = j;
    p->tail = j;
    pthread_cond_signal(&p->cv);
    pthread_mutex_unlock(&p->mu);
}

void gcn10_pool_destroy(gcn10_pool *p)
{
    if (!p)
        return;
    pthread_mutex_lock(&p->mu);
    p->stop = true;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n; i++)
        pthread_join(p->threads[i], NULL);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->cv);
    free(p->threads);
    free(p);
}

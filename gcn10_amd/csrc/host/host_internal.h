/* host_internal.h -- declarations shared by the host C files, not part of the ABI. */
#ifndef GCN10_HOST_INTERNAL_H
#define GCN10_HOST_INTERNAL_H

#include "gcn10_gpu.h"
#include "gcn10_host.h"

struct gcn10_tiff;      /* tiff.c: one open TIFF file */
struct gcn10_tiff *gcn10_tiff_open_reader(const char *path, char *err, size_t errcap);
void gcn10_tiff_close_reader(struct gcn10_tiff *t);
void gcn10_tiff_reader_info(const struct gcn10_tiff *t, int *xsize, int *ysize, double gt[6]);
const gcn10_georef *gcn10_tiff_reader_georef(const struct gcn10_tiff *t);
int gcn10_tiff_read_window(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount,
                           uint8_t *dst, size_t dst_stride, char *err, size_t errcap);


/* pool.c: fixed thread pool (tile compression, tile decode) */
typedef struct gcn10_pool gcn10_pool;
typedef void (*gcn10_job_fn)(void *arg);
gcn10_pool *gcn10_pool_create(int n_threads);
void gcn10_pool_submit(gcn10_pool *p, gcn10_job_fn fn, void *arg);
void gcn10_pool_destroy(gcn10_pool *p);     /* drains the queue first */
double gcn10_pool_cpu_seconds(gcn10_pool *p);   /* CPU time its (live) threads have used so far */
double gcn10_pool_cpu_of(gcn10_pool *p, gcn10_job_fn fn, long *n_jobs);     /* ... in jobs of one kind */
double gcn10_thread_cpu_seconds(void);          /* ... the calling thread */
void gcn10_tiff_cache_stats(uint64_t *hits, uint64_t *misses, size_t *bytes);     /* decoded-chunk cache of tiff.c */

/* Like gcn10_tiff_read_window / gcn10_raster_read, with the tiles or strips of the
 * window decoded concurrently on `pool` (NULL = on the calling thread). */
int gcn10_tiff_read_window_mt(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount,
                              uint8_t *dst, size_t dst_stride, gcn10_pool *pool, char *err,
                              size_t errcap);
int gcn10_raster_read_mt(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount, uint8_t *dst,
                         gcn10_pool *pool, char *err, size_t errcap);

/* Read plans: the compressed chunks (tiles or strips) of a window, for decoding on the GPU
 * (gcn10_gpu_inflate_tiles) instead of on the I/O pool. */
struct gcn10_chunk_ref {
    int fd;                     /* file that holds the chunk (stays open while the plan lives) */
    uint64_t file_off;
    uint32_t nbytes;            /* compressed size */
    uint32_t chunk_w, rows;     /* decoded shape */
    uint32_t src_x, src_y;      /* first wanted pixel of the chunk */
    uint32_t copy_w, copy_h;
    uint32_t dst_x, dst_y;      /* where it goes in the window */
    uint32_t flags;             /* GCN10_TILE_RAW | GCN10_TILE_PREDICTOR2 (gcn10_inflate_tile.flags) */
    uint32_t out_len;           /* bytes the chunk stands for on the device: decoded size, or nbytes of a raw one */
};
struct gcn10_read_plan {
    struct gcn10_chunk_ref *chunks;
    size_t n, cap;
    struct gcn10_tiff **opened; /* VRT sources opened for this plan */
    int n_opened;
    uint64_t covered;           /* pixels of the window the chunks fill; the rest reads as 0 */
    uint32_t max_chunk_bytes;   /* largest decoded DEFLATE chunk (raw chunks need no decode slot) */
    uint64_t staged_bytes;      /* bytes of all chunks as they cross PCIe */
};
/* 0 = planned; 1 = this window is left to the host reader (LZW / PackBits, overlapping mosaic
 * sources, full-width raw strips of a much wider raster ...): use gcn10_raster_read_mt; -1 = error (err is set). */
int gcn10_raster_plan_window(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount,
                             struct gcn10_read_plan *plan, char *err, size_t errcap);
int gcn10_tiff_plan_window(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount, int dst_x,
                           int dst_y, struct gcn10_read_plan *plan, char *err, size_t errcap);
void gcn10_read_plan_free(struct gcn10_read_plan *plan);

/* gpuapi.c: include/gcn10_gpu.h bound with dlopen */
struct gcn10_gpu_api {
    bool loaded;
    int (*abi_version)(void);
    int (*device_count)(void);
    int (*init)(int, gcn10_gpu_ctx **);
    void (*destroy)(gcn10_gpu_ctx *);
    const char *(*last_error)(void);
    int (*device_info)(gcn10_gpu_ctx *, char *, size_t, size_t *);
    int (*malloc)(gcn10_gpu_ctx *, size_t, void **);
    int (*free)(gcn10_gpu_ctx *, void *);
    int (*host_alloc)(gcn10_gpu_ctx *, size_t, void **);
    int (*host_free)(gcn10_gpu_ctx *, void *);
    int (*memcpy_h2d)(gcn10_gpu_ctx *, void *, const void *, size_t, gcn10_stream_t);
    int (*memcpy_d2h)(gcn10_gpu_ctx *, void *, const void *, size_t, gcn10_stream_t);
    int (*memset)(gcn10_gpu_ctx *, void *, int, size_t, gcn10_stream_t);
    int (*stream_create)(gcn10_gpu_ctx *, gcn10_stream_t *);
    int (*stream_destroy)(gcn10_gpu_ctx *, gcn10_stream_t);
    int (*stream_sync)(gcn10_gpu_ctx *, gcn10_stream_t);
    int (*device_sync)(gcn10_gpu_ctx *);
    int (*event_create)(gcn10_gpu_ctx *, gcn10_event_t *);
    int (*event_destroy)(gcn10_gpu_ctx *, gcn10_event_t);
    int (*event_record)(gcn10_gpu_ctx *, gcn10_event_t, gcn10_stream_t);
    int (*event_sync)(gcn10_gpu_ctx *, gcn10_event_t);
    int (*stream_wait_event)(gcn10_gpu_ctx *, gcn10_stream_t, gcn10_event_t);
    int (*event_elapsed_ms)(gcn10_gpu_ctx *, gcn10_event_t, gcn10_event_t, float *);
    int (*set_tables)(gcn10_gpu_ctx *, const int *, int);
    int (*prepare_tile)(gcn10_gpu_ctx *, const uint8_t *, int, int, const int32_t *, int,
                        gcn10_stream_t);
    int (*cn_strip)(gcn10_gpu_ctx *, const uint8_t *, int, int, const int32_t *, unsigned, unsigned,
                    uint8_t *const[GCN10_N_RASTERS], gcn10_stream_t);
    int (*pci_bus_id)(int, char *, size_t);
    int (*inflate_tiles)(gcn10_gpu_ctx *, const uint8_t *, const gcn10_inflate_tile *, int, uint32_t, uint8_t *,
                         size_t, uint32_t *, gcn10_stream_t);
    int (*deflate_fused_available)(gcn10_gpu_ctx *);
    int (*deflate_fused_strip)(gcn10_gpu_ctx *, const uint8_t *, int, int, const int32_t *, unsigned, unsigned,
                               uint8_t *, size_t, uint32_t *, unsigned long long *, gcn10_stream_t);
    size_t (*deflate_arena_bound)(int, int, int);
    int (*deflate_strip)(gcn10_gpu_ctx *, const uint8_t *const *, int, int, int, uint8_t *, size_t,
                         uint32_t *, unsigned long long *, gcn10_stream_t);
    int (*set_option)(gcn10_gpu_ctx *, const char *, int);     /* optional (tuning): NULL when the library has none */
};
const struct gcn10_gpu_api *gcn10_gpu_api_get(char *err, size_t errcap);


#endif

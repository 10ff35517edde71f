/* pipeline_internal.h -- the worker and run state shared by pipeline.c (block queue, strips,
 * sink) and pipeline_input.c (landcover input through the GPU decoder). */
#ifndef GCN10_PIPELINE_INTERNAL_H
#define GCN10_PIPELINE_INTERNAL_H

#include "gcn10_host.h"
#include "host_internal.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

enum { TILE = 256, MAX_NBUF = 8, DEFAULT_NBUF = 4, DEFAULT_STRIP_ROWS = 768, MAX_STRIP_ROWS = 4096 };

struct run;

/* one rotating set of strip buffers */
struct strip_buf {
    uint8_t *h_esa;                         /* pinned; these two only for blocks read on the host */
    uint8_t *d_esa;
    size_t esa_px;                          /* their capacity */
    uint8_t *h_out[GCN10_N_RASTERS];        /* pinned */
    uint8_t *d_out[GCN10_N_RASTERS];
    gcn10_event_t ev_h2d, ev_kernel, ev_d2h, ev_meta;
    /* GPU-side DEFLATE: compressed tiles of all 18 rasters of the strip */
    uint8_t *d_arena, *h_arena;             /* h_arena pinned */
    size_t arena_cap;                       /* device arena: the encoder's worst case      */
    size_t h_arena_cap;                     /* pinned arena: an eighth of it (>= 32 MB)    */
    uint8_t *h_spill;                       /* pageable stand-in when a strip needs more   */
    const uint8_t *h_tiles;                 /* where this strip's streams are: arena or spill */
    size_t h_tiles_used;                    /* ... and how many bytes of them the encoder produced */
    uint32_t *d_table, *h_table;            /* [18][tiles][2]; h_table pinned */
    unsigned long long *d_cursor, *h_cursor;
    const uint8_t **d_ptrs;                 /* device array of the 18 d_out pointers */
    /* compression jobs of the strip currently held by this buffer */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int pending;
    bool d2h_issued;
    int y0, rows;                           /* strip held */
    struct worker *owner;
};

struct worker {
    struct run *run;
    int rank;                               /* "rank" in the logs: outer_rank * n_workers + index */
    int index;                              /* worker index in this process; GPU = index % n_devices */
    pthread_t thread;
    gcn10_log *log;
    gcn10_gpu_ctx *ctx;
    gcn10_stream_t s_h2d, s_kernel, s_d2h;
    gcn10_raster *esa, *soil;
    size_t buf_px;                          /* capacity of one strip buffer, pixels */
    size_t buf_tiles;                       /* ... and in 256x256 tiles */
    int strip_rows;                         /* rows per strip of the current block */
    int n_cus;                              /* compute units of this worker's GPU */
    struct strip_buf buf[MAX_NBUF];         /* the first run->nbuf are in use */
    uint8_t *h_coarse;                      /* pinned: soil window, index maps of the current block */
    int32_t *h_ci, *h_cj;
    size_t h_coarse_cap, h_ci_cap, h_cj_cap;
    uint8_t *d_coarse;
    size_t coarse_cap;
    int32_t *d_ci, *d_cj;
    size_t ci_cap, cj_cap;
    atomic_bool failed;                     /* a sink job of the current block failed */
    bool fused;                             /* this worker's tables allow the fused encoder */
    /* landcover decoded on the GPU (gpu_inflate): the block's compressed chunks and where they go */
    uint8_t *h_comp, *d_comp;               /* h_comp pinned */
    size_t h_comp_cap, d_comp_cap;
    gcn10_inflate_tile *h_jobs, *d_jobs;    /* h_jobs pinned */
    uint32_t *h_status, *d_status;          /* h_status pinned */
    size_t jobs_cap;
    uint8_t *d_block;                       /* the decoded landcover block, W x H */
    size_t block_cap;
    gcn10_event_t ev_comp, ev_inflate;
    size_t n_inflate;                       /* chunks of the block in flight */
    int blocks_done;
    int device;                             /* the GPU this worker drives */
    int numa_node;                          /* ... and the NUMA node it hangs off (-1 = unknown) */
    char pci_bus[64];
    double busy_seconds;
    double t_first_block;                   /* when this worker started its first block */
    double t_read, t_gpu_wait, t_sink_wait;            /* where the worker thread's time goes */
    double t_soil, t_create, t_finish, t_device;
};

struct run {
    gcn10_config cfg;
    gcn10_run_options opt;
    const struct gcn10_gpu_api *gpu;
    gcn10_blocks blocks;
    int *block_ids;
    int n_blocks;
    int tables[9][256][5];
    int n_workers;
    struct worker *workers;
    gcn10_pool *pool;
    atomic_int next_block;
    atomic_int fatal;                       /* a worker hit an MPI_Abort-class error */
    int strip_rows;                         /* "strip_rows" of the config, rounded up to whole tile rows */
    int nbuf;                               /* strip buffer sets per worker (GCN10_STRIP_BUFFERS, 2..8) */
    int deflate_level;
    bool null_sink;                         /* GCN10_SINK=null: no compression, no files */
    bool gpu_deflate;                       /* tiles are encoded on the GPU */
    bool fused;                             /* ... straight from landcover + soil (no CN rasters in HBM) */
    bool gpu_inflate;                       /* DEFLATE landcover tiles are decoded on the GPU */
    int n_devices;                          /* visible GPUs; worker i uses device i % n_devices */
    int outer_rank, outer_size;             /* this process among the processes of an mpirun / srun */
    unsigned cond_mask, table_mask;         /* the rasters this run produces ("conditions" / "lookups") */
    int n_sel;                              /* how many: popcount(cond_mask) * popcount(table_mask) */
    int sel[GCN10_N_RASTERS];               /* their raster indices cond*9 + hc*3 + arc, ascending */
};

double gcn10_now_seconds(void);
void gcn10_wlog(struct worker *w, const char *level, bool console, const char *fmt, ...)
    __attribute__((format(printf, 4, 5)));
/* device buffer of at least `need` bytes (grown by reallocation); -1 and a log line on failure */
int gcn10_ensure_dev(struct worker *w, void **p, size_t *cap, size_t need);

/* pipeline_input.c: landcover window of a block -> w->d_block through the GPU decoder.
 * 0 = issued on s_kernel (statuses arrive with ev_inflate), 1 = this window needs the host
 * reader, -1 = the window cannot be read or decoded (logged; the block is skipped as after a failed
 * load_raster, src/cn.c:188-192), -2 = device error (logged; fatal for the run). */
int gcn10_inflate_block(struct worker *w, int xoff, int yoff, int W, int H, int block_id);

#endif

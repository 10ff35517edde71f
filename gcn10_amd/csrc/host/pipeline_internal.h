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

enum { TILE = 256, MAX_NBUF = 8, DEFAULT_NBUF = 4, DEFAULT_STRIP_ROWS = 2304, MAX_STRIP_ROWS = 4096 };

struct run;

/* one rotating set of strip buffers */
struct strip_buf {
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

/* One landcover block on its way from the files to the encoder.  Filled by the worker's input thread
 * (pipeline_input.c) while the worker encodes the block before it: window arithmetic, soil window and index
 * maps, the landcover window staged through pinned memory and decoded / untiled into d_block.  Everything in
 * it belongs to the input thread while state == FILLING and to the worker while READY. */
enum { IN_FREE = 0, IN_FILLING = 1, IN_READY = 2, IN_END = 3 };
enum { N_IN = 2 };                          /* the block being encoded + the one being staged */
enum { N_RING = 3 };                        /* pinned staging buffers of the input thread */

struct block_in {
    int state;                              /* IN_*; guarded by worker.in_mu */
    int block_id;
    int outcome;                            /* 0 = encode it; 1 = skipped (logged, like a failed load_raster,
                                               src/cn.c:188-203); -1 = an error the reference answers with MPI_Abort */
    int xoff, yoff, W, H, hsx, hsy;
    double gt[6];
    uint8_t *h_coarse;                      /* pinned: soil window, index maps */
    int32_t *h_ci, *h_cj;
    size_t h_coarse_cap, h_ci_cap, h_cj_cap;
    uint8_t *d_coarse;
    int32_t *d_ci, *d_cj;
    size_t coarse_cap, ci_cap, cj_cap;
    uint8_t *d_block;                       /* the landcover window, W x H, row major */
    size_t block_cap;
    uint8_t *d_comp;                        /* the window's chunks as they lie in the files (compressed or raw) */
    size_t d_comp_cap;
    gcn10_inflate_tile *h_jobs, *d_jobs;    /* h_jobs pinned */
    uint32_t *h_status, *d_status;          /* h_status pinned */
    size_t jobs_cap;
    size_t n_inflate;                       /* chunks whose status must be looked at (0: host reader) */
    gcn10_event_t ev_ready;                 /* recorded behind everything the block needs on the device */
    /* the block's output files (src/cn.c:236-360: directories, names, the trailing-underscore rule), created by
     * the input side while the block before is encoded: 18 open + truncate calls are 5-10 ms per block */
    gcn10_tiff_writer *tifs[GCN10_N_RASTERS];
    bool tifs_ok;                           /* all of the run's rasters have their file */
};

struct worker {
    struct run *run;
    int rank;                               /* "rank" in the logs: outer_rank * n_workers + index */
    int index;                              /* worker index in this process; GPU = index % n_devices */
    pthread_t thread;
    gcn10_log *log;
    gcn10_gpu_ctx *ctx;
    gcn10_stream_t s_kernel, s_d2h;
    gcn10_raster *esa, *soil;
    size_t buf_px;                          /* capacity of one strip buffer, pixels */
    size_t buf_tiles;                       /* ... and in 256x256 tiles */
    int strip_rows;                         /* rows per strip of the current block */
    int n_cus;                              /* compute units of this worker's GPU */
    struct strip_buf buf[MAX_NBUF];         /* the first run->nbuf are in use */
    atomic_bool failed;                     /* a sink job of the current block failed */
    bool fused;                             /* this worker's tables allow the fused encoder */
    /* input side: a thread of its own (prefetch_blocks=1) or the worker itself, in turn */
    pthread_t in_thread;
    bool in_thread_started;
    gcn10_gpu_ctx *in_ctx;                  /* the input thread's own context on the same device */
    gcn10_stream_t s_in;
    struct block_in in[N_IN];
    pthread_mutex_t in_mu;
    pthread_cond_t in_cv;
    bool in_stop;                           /* the worker is going away: the input thread must not wait for it */
    uint8_t *h_ring[N_RING];                /* pinned: chunks (or host-decoded rows) on their way to the device */
    size_t ring_cap;
    gcn10_event_t ev_ring[N_RING];
    bool ring_busy[N_RING];
    int blocks_done;
    int in_seq;                             /* blocks taken from the input side so far: slot = in_seq % N_IN */
    int device;                             /* the GPU this worker drives */
    int numa_node;                          /* ... and the NUMA node it hangs off (-1 = unknown) */
    char pci_bus[64];
    double busy_seconds;
    double cpu_seconds, in_cpu_seconds;        /* CPU time of the worker's thread and of its input thread */
    double t_first_block;                   /* when this worker started its first block */
    double t_first_done, t_last_done;       /* when it finished its first / its last block */
    double t_gpu_wait, t_sink_wait;         /* where the worker thread's time goes */
    double t_create, t_finish, t_device, t_in_wait;
    double t_read, t_soil, t_in_busy;       /* ... and the input thread's */
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
    atomic_llong pinned_bytes;              /* pinned host memory asked for by all workers (allocations; regrowth counts twice) */
    atomic_int next_block;
    atomic_int fatal;                       /* a worker hit an MPI_Abort-class error */
    int strip_rows;                         /* "strip_rows" of the config, rounded up to whole tile rows */
    int nbuf;                               /* strip buffer sets per worker (GCN10_STRIP_BUFFERS, 2..8) */
    int event_sleep_us;                     /* GPU events are waited for with query + sleep (0: the runtime's spinning wait) */
    int drain_lag;                          /* the strip handed to the sink while strip s is submitted: s - drain_lag */
    int deflate_level;
    bool null_sink;                         /* GCN10_SINK=null: no compression, no files */
    bool gpu_deflate;                       /* tiles are encoded on the GPU */
    bool fused;                             /* ... straight from landcover + soil (no CN rasters in HBM) */
    bool gpu_inflate;                       /* DEFLATE landcover tiles are decoded on the GPU */
    bool direct_io;                         /* tile data is written with O_DIRECT */
    bool prefetch;                          /* input threads stage block N+1 while block N is encoded */
    int n_devices;                          /* GPUs of the run; worker i belongs to GPU i % n_devices */
    int n_physical;                         /* ... and the devices behind them: GPU d is device d % n_physical (all the
                                               same unless GCN10_REHEARSE_GPUS rehearses a bigger node on this one) */
    int outer_rank, outer_size;             /* this process among the processes of an mpirun / srun */
    unsigned cond_mask, table_mask;         /* the rasters this run produces ("conditions" / "lookups") */
    int n_sel;                              /* how many: popcount(cond_mask) * popcount(table_mask) */
    int sel[GCN10_N_RASTERS];               /* their raster indices cond*9 + hc*3 + arc, ascending */
};

double gcn10_now_seconds(void);
void gcn10_wlog(struct worker *w, const char *level, bool console, const char *fmt, ...)
    __attribute__((format(printf, 4, 5)));

/* device / pinned buffer of at least `need` bytes on context `ctx` (grown by reallocation); -1 and a log line on failure */
int gcn10_ensure_dev_on(struct worker *w, gcn10_gpu_ctx *ctx, void **p, size_t *cap, size_t need);
int gcn10_ensure_pinned_on(struct worker *w, gcn10_gpu_ctx *ctx, void **p, size_t *cap, size_t need);

/* pipeline_input.c: the input side of a worker.
 * gcn10_input_setup / _teardown: context, stream, events, pinned ring of the input side.
 * gcn10_input_start: starts the input thread (prefetch) -- it takes block ids from the run's counter, fills
 *   w->in[] slots in turn and marks them IN_READY (IN_END after the last one).
 * gcn10_input_next: the slot the worker encodes next (waits for it; without a thread, fills it right here);
 *   NULL at the end of the queue.  gcn10_input_release hands the slot back. */
/* pipeline.c: the output files of a block (directories, names, overwrite rule); 0, 1 = a file could not be created
 * (logged like save_raster logs, the block is given up), -1 = an output directory cannot be made (fatal, src/cn.c:250) */
int gcn10_create_outputs(struct worker *w, struct block_in *in);
void gcn10_abort_outputs(struct block_in *in);
int gcn10_input_setup(struct worker *w);
void gcn10_input_teardown(struct worker *w);
int gcn10_input_start(struct worker *w);
struct block_in *gcn10_input_next(struct worker *w);
void gcn10_input_release(struct worker *w, struct block_in *in);
void gcn10_input_stop(struct worker *w);

#endif

/*
 * gcn10_gpu.h -- C ABI of the MI355X (gfx950) curve-number engine.
 *
 * This is the drop-in boundary for gcn10's per-pixel path.  The reference has
 * no FFI layer; its seam is the set of C prototypes in src/global.h:47-71 and
 * the static kernels of src/cn.c.  Each entry point below names the reference
 * code it replaces.  Plain C types only: no HIP or torch types cross this line.
 * A stream or event is an opaque pointer (a hipStream_t / hipEvent_t inside);
 * passing NULL as a stream means "the context's own main stream".
 *
 * All functions returning int give 0 on success or a negative GCN10_E_* code;
 * gcn10_gpu_last_error() returns the calling thread's last message.  The
 * reference's functions are void and log-then-return or MPI_Abort
 * (src/cn.c:156-160, 25); the host driver maps these codes back to that
 * behaviour.
 *
 * Threading: a context belongs to one GPU and is used by one host thread at a
 * time (the reference runs one thread per rank, src/main.c:171-176).  The one
 * exception: gcn10_gpu_event_sync() only reads the context and may be called by
 * other threads while its owner works (the host pipeline's I/O threads wait for
 * a strip's copy-back that way).
 *
 * There is no CPU fallback behind this ABI: every call needs a live gfx950
 * device, and gcn10_gpu_init() fails loudly without one.
 */
#ifndef GCN10_GPU_H
#define GCN10_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: + gcn10_gpu_tune_single_raster, gcn10_gpu_stream_copy, gcn10_gpu_soil_words_state (round 2).
 * 3: gcn10_inflate_tile.reserved became .flags (raw and predictor-2 chunks), the tile encoders lay a raster's
 *    streams of a strip out as one extent (option "arena_segment_align"), gcn10_gpu_deflate_arena_bound grew by
 *    the extents' pads (round 3).  A caller built against an older header must be rebuilt: check at start-up. */
#define GCN10_GPU_ABI_VERSION 3

enum {
    GCN10_OK = 0,
    GCN10_E_INVAL = -1,     /* bad argument (null pointer, size, mask)        */
    GCN10_E_HIP = -2,       /* a HIP runtime call failed; see last_error      */
    GCN10_E_NOMEM = -3,     /* device or pinned-host allocation failed        */
    GCN10_E_STATE = -4,     /* call order: tables / tile not prepared         */
    GCN10_E_NODEVICE = -5   /* no gfx950 device visible                       */
};

/* Raster order of the 18 outputs of one block: index = cond*9 + hc*3 + arc,
 * cond in {drained, undrained}, hc in {p, f, g}, arc in {i, ii, iii}
 * -- the loop order of src/cn.c:145-147, 236-259. */
#define GCN10_N_TABLES 9
#define GCN10_N_CONDS 2
#define GCN10_N_RASTERS 18
#define GCN10_COND_DRAINED 1u       /* cond_mask bit for conds[0] (src/cn.c:145) */
#define GCN10_COND_UNDRAINED 2u     /* cond_mask bit for conds[1]                */
#define GCN10_NODATA 255            /* src/cn.c:38, 289                          */

typedef struct gcn10_gpu_ctx gcn10_gpu_ctx;
typedef void *gcn10_stream_t;
typedef void *gcn10_event_t;

/* ---- life cycle -------------------------------------------------------- */

int gcn10_gpu_abi_version(void);
/* Number of visible HIP devices (0 when there is none or the runtime fails). */
int gcn10_gpu_device_count(void);
/* One context per GPU.  Replaces the per-rank process state of the reference
 * (MPI_Init + rank, src/main.c:80-82): "rank" becomes the device index. */
int gcn10_gpu_init(int device, gcn10_gpu_ctx **ctx);
void gcn10_gpu_destroy(gcn10_gpu_ctx *ctx);
const char *gcn10_gpu_last_error(void);
/* Fills name[cap] with the device name and returns CU count, or <0. */
int gcn10_gpu_device_info(gcn10_gpu_ctx *ctx, char *name, size_t cap,
                          size_t *hbm_bytes);

/* PCI address of HIP device `device` as "dddd:bb:dd.f" (for NUMA placement of the
 * host thread that feeds it); works without a context.  0 or <0. */
int gcn10_gpu_pci_bus_id(int device, char *buf, size_t cap);

/* ---- memory, streams, events ------------------------------------------ */
/* The reference mallocs every raster on the host (src/raster.c:169,
 * src/cn.c:209,264,278).  Here rasters live in HBM; the host stages them
 * through pinned buffers (replaces GDALRasterIO's destination buffer,
 * src/raster.c:176-178, 217-219). */

int gcn10_gpu_malloc(gcn10_gpu_ctx *ctx, size_t bytes, void **dptr);
int gcn10_gpu_free(gcn10_gpu_ctx *ctx, void *dptr);
int gcn10_gpu_host_alloc(gcn10_gpu_ctx *ctx, size_t bytes, void **hptr);
int gcn10_gpu_host_free(gcn10_gpu_ctx *ctx, void *hptr);
/* Asynchronous on `stream`; the host buffer must be pinned for real overlap. */
int gcn10_gpu_memcpy_h2d(gcn10_gpu_ctx *ctx, void *dst_dev, const void *src_host,
                         size_t bytes, gcn10_stream_t stream);
int gcn10_gpu_memcpy_d2h(gcn10_gpu_ctx *ctx, void *dst_host, const void *src_dev,
                         size_t bytes, gcn10_stream_t stream);
int gcn10_gpu_memset(gcn10_gpu_ctx *ctx, void *dptr, int value, size_t bytes,
                     gcn10_stream_t stream);

int gcn10_gpu_stream_create(gcn10_gpu_ctx *ctx, gcn10_stream_t *stream);
int gcn10_gpu_stream_destroy(gcn10_gpu_ctx *ctx, gcn10_stream_t stream);
int gcn10_gpu_stream_sync(gcn10_gpu_ctx *ctx, gcn10_stream_t stream);
int gcn10_gpu_device_sync(gcn10_gpu_ctx *ctx);
int gcn10_gpu_event_create(gcn10_gpu_ctx *ctx, gcn10_event_t *ev);
int gcn10_gpu_event_destroy(gcn10_gpu_ctx *ctx, gcn10_event_t ev);
int gcn10_gpu_event_record(gcn10_gpu_ctx *ctx, gcn10_event_t ev, gcn10_stream_t stream);
int gcn10_gpu_event_sync(gcn10_gpu_ctx *ctx, gcn10_event_t ev);
/* Makes `stream` wait for `ev` (device-side; the host does not block). */
int gcn10_gpu_stream_wait_event(gcn10_gpu_ctx *ctx, gcn10_stream_t stream, gcn10_event_t ev);
int gcn10_gpu_event_elapsed_ms(gcn10_gpu_ctx *ctx, gcn10_event_t start,
                               gcn10_event_t stop, float *ms);

/* ---- lookup tables ----------------------------------------------------- */
/* Replaces the hand-off `int table[256][5]` from load_lookup_table() to
 * calculate_cn() (src/cn.c:148, 261, 290).  `tables` is n_tables consecutive
 * reference-format tables, i.e. int[n_tables][256][5] exactly as
 * load_lookup_table fills them (255 = nodata).  The engine folds the
 * `cn_value < 255 ? (uint8_t)cn_value : keep 255` rule of src/cn.c:125-128
 * into byte tables once, and keeps them on the device until replaced.
 * 1 <= n_tables <= 9. */
int gcn10_gpu_set_tables(gcn10_gpu_ctx *ctx, const int *tables, int n_tables);

/* ---- per-function kernels (one reference function each) ---------------- */

/* src/cn.c:218-232, the resample loop, with its separable index arithmetic
 * hoisted to the host: ci[W] and cj[rows] are the clamped coarse column / row
 * of every fine column / row (gcn10_build_index_maps in gcn10_host.h computes
 * them in the reference's exact fp64 order).  out[y*W+x] =
 * coarse[cj[y]*hsx+ci[x]].  All pointers are device pointers. */
int gcn10_gpu_resample(gcn10_gpu_ctx *ctx, const uint8_t *coarse, int hsx, int hsy,
                       const int32_t *ci, const int32_t *cj, int W, int rows,
                       uint8_t *out, gcn10_stream_t stream);

/* src/cn.c:88-111 modify_hysogs_data(), in place on a device buffer;
 * drained != 0 <=> cond == "drained". */
int gcn10_gpu_modify_hysogs_data(gcn10_gpu_ctx *ctx, uint8_t *h, size_t npix,
                                 int drained, gcn10_stream_t stream);

/* src/cn.c:114-131 calculate_cn() on a raster the reference pre-fills with 255
 * (src/cn.c:289): out[i] = (hsg[i] < 5 && T[esa[i]][hsg[i]] < 255) ?
 * (uint8_t)T[esa[i]][hsg[i]] : 255, T = table `table_index` of set_tables. */
int gcn10_gpu_calculate_cn(gcn10_gpu_ctx *ctx, const uint8_t *esa,
                           const uint8_t *hsg, size_t npix, int table_index,
                           uint8_t *out, gcn10_stream_t stream);

/* ---- fused block path --------------------------------------------------- */
/* Replaces the body of process_block() between load_raster and save_raster
 * (src/cn.c:205-290): resample + memcpy + modify_hysogs_data + memset +
 * calculate_cn for every (cond, hc, arc), in one pass over the landcover.
 *
 * gcn10_gpu_prepare_tile: once per block.  Expands the coarse soil window
 * along x only (one row of W bytes per coarse row, kept in a device workspace
 * that stays L2 / Infinity-Cache resident) -- the x half of src/cn.c:218-232.
 * Next to the bytes it writes one compact word per 16-pixel column group and
 * coarse row (two codes and the position where the second begins); strips
 * whose width is a multiple of 16 read those instead of the bytes when every
 * group of the tile has that form (gcn10_gpu_soil_words_state).
 *
 * gcn10_gpu_cn_strip: any number of times per block, one row strip each
 * (rows y0 .. y0+rows of the block; `esa` and every out[] pointer address the
 * first pixel of the strip; cj points at the strip's first entry, i.e.
 * &cj_block[y0]).  For every selected raster r = cond*9 + k it writes
 *     h = coarse[cj[y]*hsx + ci[x]];  s = remap_cond(h)   (src/cn.c:88-111)
 *     out[r][y*W+x] = s < 5 ? T8[k][esa[y*W+x]][s] : 255  (src/cn.c:114-131)
 * cond_mask: GCN10_COND_* bits; table_mask: bit k selects table k (< n_tables).
 * out[] has GCN10_N_RASTERS entries (host array of device pointers); entries
 * of unselected rasters are ignored and may be NULL.  Strips of one block may
 * be issued on different streams once prepare_tile's stream work is ordered
 * before them (event or sync).
 * out[] must point into ordinary device allocations (gcn10_gpu_malloc / hipMalloc).
 * Ranges assembled with HIP's virtual memory management (hipMemCreate + hipMemMap)
 * are not supported: the kernels store nontemporally, and such stores into chunk-
 * mapped ranges were observed not to be visible yet when the kernel had completed
 * (ROCm 7.2, MI355X; profiles/r02/spread_allocator_hazard.txt). */
int gcn10_gpu_prepare_tile(gcn10_gpu_ctx *ctx, const uint8_t *coarse, int hsx,
                           int hsy, const int32_t *ci, int W,
                           gcn10_stream_t stream);
int gcn10_gpu_cn_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows,
                       const int32_t *cj, unsigned cond_mask, unsigned table_mask,
                       uint8_t *const out[GCN10_N_RASTERS], gcn10_stream_t stream);

/* Bytes the dominant kernel (cn_strip) must move per launch by the algorithm:
 * W*rows*(1 + n_selected_rasters) + the x-expanded soil rows it reads once.
 * Used by the benchmark for the roofline figure ("algorithmic
 * bytes"). */
size_t gcn10_gpu_strip_algorithmic_bytes(int W, int rows, int hsx, int hsy,
                                         unsigned cond_mask, unsigned table_mask);

/* ---- output encode: the DEFLATE of save_raster(), on the GPU --------------- */
/* Replaces the zlib work inside save_raster()'s GDALRasterIO (src/raster.c:204-219:
 * GTiff, COMPRESS=DEFLATE, TILED=YES, i.e. one zlib stream per 256x256 block).
 * Every 256x256 tile of `n_rasters` device raster strips (W x rows each, row
 * major; edge tiles zero padded) becomes one complete zlib stream (RFC 1950/1951,
 * dynamic Huffman) in `arena_dev`; table_dev[(r*tiles + ty*across + tx)*2 + {0,1}]
 * receives the stream's byte offset in the arena and its size (offset 0xffffffff:
 * the arena was too small); *cursor_dev receives the number of arena bytes used.
 * rasters_dev is a DEVICE array of n_rasters device pointers.  Asynchronous on
 * `stream`.  The raw rasters never have to cross PCIe. */
size_t gcn10_gpu_deflate_arena_bound(int W, int rows, int n_rasters);
int gcn10_gpu_deflate_strip(gcn10_gpu_ctx *ctx, const uint8_t *const *rasters_dev,
                            int n_rasters, int W, int rows, uint8_t *arena_dev,
                            size_t arena_cap, uint32_t *table_dev,
                            unsigned long long *cursor_dev, gcn10_stream_t stream);

/* The same encoding WITHOUT materialising the CN rasters: the strip's landcover and the
 * block's x-expanded soil (gcn10_gpu_prepare_tile) are reduced to pixel classes once per
 * tile position, tokenised once, and the selected rasters' streams are produced from that
 * (src/cn.c:236-290 and the zlib of src/raster.c:204-219 in one device pass; no CN raster
 * is written to or read from HBM).  Stream r' of the table is the r'-th selected raster in
 * ascending raster order (cond*9 + hc*3 + arc).  Needs the tables to define at most 256
 * distinct 18-vectors (GCN10_E_STATE otherwise; the shipped tables define about 110).
 * Two table entries may name the same arena bytes: where no dual soil class lies under a tile,
 * the drained and the undrained raster of a table are the same stream, emitted once. */
int gcn10_gpu_deflate_fused_available(gcn10_gpu_ctx *ctx);     /* 1 after set_tables if <= 256 classes */
int gcn10_gpu_deflate_fused_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows,
                                  const int32_t *cj, unsigned cond_mask, unsigned table_mask,
                                  uint8_t *arena_dev, size_t arena_cap, uint32_t *table_dev,
                                  unsigned long long *cursor_dev, gcn10_stream_t stream);

/* Landcover input decode on the GPU (replaces the inflate GDALRasterIO does on the host inside
 * load_raster, /root/reference/src/raster.c:167-176).  `tiles_dev` describes n_tiles zlib
 * streams (TIFF Compression 8 / 32946: one per tile or strip of the landcover files) lying in
 * `comp_dev`; every stream is decoded by one workgroup (a decoder and a copier wavefront) into
 * a linear slot of at most chunk_bytes, and the wanted window of each chunk (copy_w x copy_h pixels from (src_x, src_y)
 * of a chunk chunk_w pixels wide) is copied to dst_dev + dst_off, rows dst_stride apart.
 * status_dev[i] = 0, or the reason stream i is not a valid zlib stream (GCN10_INFLATE_E_*); a
 * stream that ends early leaves zeros, one that is longer than out_len is cut there, as the
 * host reader (tiff.c) does.  Streams must start at multiples of 16 bytes in comp_dev, and
 * comp_dev must be readable up to the next multiple of 4 past every stream's end. */
typedef struct gcn10_inflate_tile {
    uint64_t in_off;        /* byte offset of the zlib stream in comp_dev, multiple of 16 */
    uint32_t in_len;        /* its size in bytes */
    uint32_t out_len;       /* bytes the chunk decodes to: chunk_w * rows in the chunk */
    uint32_t chunk_w;       /* pixels per row of the decoded chunk */
    uint32_t src_x, src_y;  /* first wanted pixel of the chunk */
    uint32_t copy_w, copy_h;
    uint32_t flags;         /* GCN10_TILE_* (ABI 3; 0 = a zlib stream, no predictor, as in ABI 1-2) */
    uint64_t dst_off;       /* where pixel (src_x, src_y) goes in dst_dev */
} gcn10_inflate_tile;
/* flags: RAW = the chunk is not compressed (TIFF Compression 1): in_len >= out_len bytes of pixels lie at
 * in_off and only the window copy is done (no alignment rule for in_off; the host may stage only the bytes
 * from the first wanted pixel on: src_x = src_y = 0 then, chunk_w still the chunk's row pitch).
 * PREDICTOR2 = the chunk was written with TIFF Predictor 2 (horizontal differencing): every chunk row is
 * summed back, byte-wise modulo 256, from the row's first pixel while it is copied (tiff.c decode_chunk does
 * the same on the host). */
#define GCN10_TILE_RAW 1u
#define GCN10_TILE_PREDICTOR2 2u
enum {
    GCN10_INFLATE_E_HEADER = 1, GCN10_INFLATE_E_BLOCK_TYPE = 2, GCN10_INFLATE_E_STORED = 3,
    GCN10_INFLATE_E_LENGTHS = 4, GCN10_INFLATE_E_CODE = 5, GCN10_INFLATE_E_DISTANCE = 6,
    GCN10_INFLATE_E_INPUT = 7, GCN10_INFLATE_E_WINDOW = 8
};
int gcn10_gpu_inflate_tiles(gcn10_gpu_ctx *ctx, const uint8_t *comp_dev,
                            const gcn10_inflate_tile *tiles_dev, int n_tiles, uint32_t chunk_bytes,
                            uint8_t *dst_dev, size_t dst_stride, uint32_t *status_dev,
                            gcn10_stream_t stream);

/* Launch-shape knobs of the strip kernels, for tuning runs; results never
 * depend on them.  Names: "grid_blocks_per_cu" (1..64), "ilp16" (0 = by raster count | 1 | 2),
 * "ilp1" (1|2|4), "nontemporal" (0|1), "xcd_slabs" (0|1), "prefetch" (software pipeline of the strip kernels: -1 = default = on | 0 | 1), "compact_soil" (0|1: gcn10_gpu_prepare_tile writes,
 * and strips of 16-byte aligned rows read, the compact soil words; default 1), "deflate_wave_codes" (0|1: code
 * construction of the tile encoder by one thread or one wave per tile), "fused_diag" (timing
 * experiments on the fused encoder; nonzero values produce invalid streams), "defaults" (value
 * ignored: every knob back to its built-in default).
 * Round 3: "arena_segment_align" (16..4096, a power of two; default 4096: every raster's extent of a strip starts at a
 * multiple of it in the arena), "fused_parse" / "fused_emit" (0 = the round-2 forms of the fused encoder's first and
 * last pass, kept as cross-checks; 1 = default; the streams are the same bytes), "fused_stats_stop", "codes_stop",
 * "inflate_diag" (timing experiments, like "fused_diag"), "event_sync_sleep_us" (0 = default: gcn10_gpu_event_sync
 * is hipEventSynchronize, which spins; n > 0: it queries the event and sleeps n microseconds in between -- what a
 * host pipeline's waiting threads want). */
int gcn10_gpu_set_option(gcn10_gpu_ctx *ctx, const char *name, int value);

/* Measurement: the next gcn10_gpu_cn_strip launch records `start` / `stop` as part of
 * the kernel dispatch itself (hipExtLaunchKernel), so gcn10_gpu_event_elapsed_ms gives
 * the kernel's own duration -- the figure rocprofv3 --kernel-trace reports. One shot. */
int gcn10_gpu_time_next_strip(gcn10_gpu_ctx *ctx, gcn10_event_t start, gcn10_event_t stop);

/* Measurement: a plain copy of `bytes` (a multiple of 16, both pointers 16-byte aligned)
 * from `src` to `dst`, with the strip kernel's launch shape and nothing but the two
 * streams: what this device sustains for 1 byte read : 1 byte written.  bench.py times
 * it next to the strip kernel, in the same run.  Honours gcn10_gpu_time_next_strip.
 * Kernel name in profiles: stream_copy_kernel.  No counterpart in the reference. */
int gcn10_gpu_stream_copy(gcn10_gpu_ctx *ctx, const void *src, void *dst, size_t bytes,
                          gcn10_stream_t stream);

/* Placement and launch-shape calibration for one-raster strips (BASELINE config 2 / config 3).
 * A strip with one raster is a 1 byte read : 1 byte written stream, and on MI355X the rate of such a
 * stream depends on where the written buffer lies relative to the read one -- periodically in the
 * distance, with a 128 MiB period for a grid-stride sweep, by up to 20 % (DESIGN.md section 5,
 * profiles/r02/offset_lab_*.jsonl) -- and the best workgroup -> address mapping depends on it too.
 * The call times the single-raster kernel of (cond_mask, table_mask: one bit each) on this very
 * landcover strip for every position arena + i * step (step a multiple of 256; while a raster of
 * W * rows bytes still fits in arena_bytes) and a fixed set of launch shapes, keeps the fastest
 * shape for later single-raster launches of this context (the all-tables kernel is not affected;
 * gcn10_gpu_set_option "ilp1" or "defaults" drops it) and returns the fastest position.
 * The arena is caller-owned device memory; its contents are overwritten.  `report` (optional)
 * receives a one-line JSON summary.  Results of later launches do not depend on any of this.
 * Synchronises `stream`.  No counterpart in the reference (its buffers are malloc'ed per raster,
 * src/cn.c:264,278). */
int gcn10_gpu_tune_single_raster(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows,
                                 const int32_t *cj, unsigned cond_mask, unsigned table_mask,
                                 uint8_t *arena, size_t arena_bytes, size_t step,
                                 uint8_t **best_out, float *best_ms, char *report, size_t report_cap,
                                 gcn10_stream_t stream);

/* What the strips of the tile last prepared on this context read their soil codes from (synchronises
 * `stream`, on which gcn10_gpu_prepare_tile ran): 0 = code bytes (compact words switched off with
 * gcn10_gpu_set_option("compact_soil", 0)), 1 = compact words -- one dword per 16-pixel column group, used by
 * strips whose width is a multiple of 16 --, 2 = code bytes because some column group of this tile spans
 * more than two soil cells.  <0 on error.  A diagnostic: nothing needs to call it. */
int gcn10_gpu_soil_words_state(gcn10_gpu_ctx *ctx, gcn10_stream_t stream);

/* Name of the variant of the strip kernel the last cn_strip call launched
 * (for profiles and bench records). */
const char *gcn10_gpu_last_kernel_name(gcn10_gpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GCN10_GPU_H */

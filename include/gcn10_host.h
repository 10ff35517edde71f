/*
 * gcn10_host.h -- host side (plain C99) of the MI355X curve-number generator.
 *
 * These are the pieces of gcn10's src/ program that stay on the CPU: the
 * lookup-CSV loader, the fp64 geotransform arithmetic that must be bit-exact,
 * config / logging / block list handling, raster I/O and the per-GPU block
 * queue that drives include/gcn10_gpu.h.  Every declaration names the
 * reference code whose behaviour it keeps (paths under /root/reference).
 */
#ifndef GCN10_HOST_H
#define GCN10_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCN10_VERSION "0.1.0"          /* src/main.c:12-14 */

/* ------------------------------------------------------------------------ */
/* lookup tables (src/cn.c:13-85)                                           */
/* ------------------------------------------------------------------------ */

/* hydrologic conditions and ARCs in the reference's loop order
 * (src/cn.c:146-147); table k = hc*3 + arc. */
extern const char *const gcn10_hcs[3];      /* "p", "f", "g"     */
extern const char *const gcn10_arcs[3];     /* "i", "ii", "iii"  */
extern const char *const gcn10_conds[2];    /* "drained", "undrained" (src/cn.c:145) */

/* Callback for the rows load_lookup_table() logs as ERROR and skips
 * (src/cn.c:58-63, 68-73, 78-82).  May be NULL. */
typedef void (*gcn10_row_error_fn)(void *user, const char *message);

/* Parses one lookup CSV into the reference's int table[256][5].
 * Returns 0, -1 when the file cannot be opened (src/cn.c:28-33) and -2 when it
 * is empty (src/cn.c:43-48); the reference aborts the run in both cases and so
 * does the gcn10 program. */
int gcn10_load_lookup_file(const char *path, int table[256][5],
                           gcn10_row_error_fn on_error, void *user);

/* "<dir>/default_lookup_<hc>_<arc>.csv" (src/cn.c:21).  -3: path too long. */
int gcn10_load_lookup_table(const char *dir, const char *hc, const char *arc,
                            int table[256][5], gcn10_row_error_fn on_error,
                            void *user);

/* All nine tables in k = hc*3+arc order; stops at the first failure and
 * returns its code (failed_k, if not NULL, receives the table index). */
int gcn10_load_all_lookup_tables(const char *dir, int tables[9][256][5],
                                 int *failed_k, gcn10_row_error_fn on_error,
                                 void *user);

/* ------------------------------------------------------------------------ */
/* geotransform arithmetic                                                   */
/* ------------------------------------------------------------------------ */

/* The separable index maps of the resample loop, src/cn.c:218-229:
 *   ci[x] = clamp((int)round((gt[0]+(x+0.5)*gt[1] - soil_gt[0]) / soil_gt[1]), 0, hsx-1)
 *   cj[y] = clamp((int)round((soil_gt[3] - (gt[3]+(y+0.5)*gt[5])) / fabs(soil_gt[5])), 0, hsy-1)
 * evaluated in IEEE double, in the reference's operation order, without
 * fused multiply-add (this file is built with -ffp-contract=off), and with the
 * reference build's x86-64 double->int conversion. */
void gcn10_build_index_maps(const double gt[6], const double soil_gt[6],
                            int W, int H, int hsx, int hsy,
                            int32_t *ci, int32_t *cj);

/* Window of a raster (geotransform t, size rx x ry) covering bbox
 * {minx, miny, maxx, maxy}: src/raster.c:126-162.  Returns 0, or -1 for the
 * "invalid raster bounds" case (src/raster.c:142-147). */
int gcn10_raster_window(const double t[6], int rx, int ry, const double bbox[4],
                        int *xoff, int *yoff, int *xcount, int *ycount,
                        double gt[6]);


/* ------------------------------------------------------------------------ */
/* config (src/config.c) and logging (src/log.c)                            */
/* ------------------------------------------------------------------------ */

/* The five required keys of the reference's config file (src/config.c:69-103,
 * 107-113) plus optional keys this program adds; unknown keys and lines
 * without '=' are ignored, '#' starts a comment line, lines are cut at 511
 * bytes (src/config.c:47-64). */
typedef struct gcn10_config {
    char *hysogs_data_path;
    char *esa_data_path;
    char *blocks_shp_path;
    char *lookup_table_path;
    char *log_dir;
    /* optional extensions (absent = default) */
    int gpus;               /* "gpus": number of GPUs to use, 0 = all visible      */
    int workers_per_gpu;    /* "workers_per_gpu": block workers per GPU, 0 = default 2 */
    int strip_rows;         /* "strip_rows": rows per staging strip, 0 = default 768 */
    int io_threads;         /* "io_threads": tile compression threads, 0 = auto    */
    int deflate_level;      /* "deflate_level": zlib level 1..9, 0 = zlib default 6 */
    char *esa_tile_dir;     /* "esa_tile_dir": local mirror of /vsicurl/ VRT sources */
    int gpu_deflate;        /* "gpu_deflate": 2 (default) fused: tiles are DEFLATE-encoded on the
                               GPU straight from landcover + soil, no CN raster in HBM;
                               1 = CN strips in HBM, then encoded on the GPU per raster;
                               0 = raw strips are copied back, host zlib threads encode */
    int gpu_inflate;        /* "gpu_inflate": 1 (default) DEFLATE-compressed landcover tiles cross PCIe
                               compressed and are decoded on the GPU, uncompressed ones are untiled there,
                               TIFF predictor 2 is undone there; 0 = all of it on the host i/o pool */
    int direct_io;          /* "direct_io": 1 = the GeoTIFFs' tile data is written with O_DIRECT from the pinned
                               copy of the encoder's arena (no page-cache copy); 0 (default) = buffered writes */
    int prefetch_blocks;    /* "prefetch_blocks": 1 (default) = every block worker has an input thread that stages
                               and decodes the NEXT block's landcover while this one is encoded; 0 = in turn */
    unsigned table_mask;    /* "lookups": which of the nine lookups to produce, e.g. "g_ii" or "p_i,f_iii"
                               ("all" / absent = all nine); bit k = hc*3 + arc in the reference's loop
                               order p,f,g x i,ii,iii (src/cn.c:146-147) */
    unsigned cond_mask;     /* "conditions": "drained", "undrained" or "both" (absent = both);
                               bit 0 = drained, bit 1 = undrained (src/cn.c:145) */
} gcn10_config;

/* "g_ii", "p_i,f_iii", "all" -> table mask; "drained" | "undrained" | "both" | "all" -> condition mask.
 * Return 0 and set *mask, or -1 for a name that is not a lookup / condition. */
int gcn10_parse_lookups(const char *text, unsigned *mask);
int gcn10_parse_conditions(const char *text, unsigned *mask);

/* Returns 0; -1 cannot open (message in err); -3 a bad "lookups" / "conditions" value; -2 a required key is missing
 * (the reference aborts in both cases, src/config.c:50-54, 107-113). */
int gcn10_config_parse(const char *path, gcn10_config *cfg, char *err, size_t errcap);
void gcn10_config_free(gcn10_config *cfg);

/* Per-worker log, "<log_dir>/rank_<r>.log", append mode, lines
 * "[%Y-%m-%dT%H:%M:%S] [LEVEL] [rank r] msg" (src/log.c:67-86, 149-166).
 * A "rank" is a GPU worker here.  Thread safe. */
typedef struct gcn10_log gcn10_log;
gcn10_log *gcn10_log_open(const char *log_dir, int rank);       /* = init_logging  */
void gcn10_log_message(gcn10_log *lg, const char *level, const char *msg,
                       bool also_console);                      /* = log_message   */
void gcn10_log_close(gcn10_log *lg);                            /* = finalize_logging */

/* ------------------------------------------------------------------------ */
/* block index (src/raster.c:23-103, src/cn.c:155-184)                      */
/* ------------------------------------------------------------------------ */

/* Whitespace-separated integers; parsing stops at the first token that is not
 * an integer (fscanf("%d"), src/raster.c:46).  Returns a malloc'd array. */
int *gcn10_read_block_list(const char *path, int *n_blocks);

/* The polygon shapefile of block extents, read without OGR: the .shp record
 * bounding boxes and the "ID" column of the .dbf. */
typedef struct gcn10_blocks {
    int n;
    int *id;                /* "ID" attribute of every feature, file order   */
    double (*bbox)[4];      /* {minx, miny, maxx, maxy} = OGR envelope       */
} gcn10_blocks;

int gcn10_blocks_open(const char *shp_path, gcn10_blocks *out, char *err, size_t errcap);
void gcn10_blocks_free(gcn10_blocks *b);
/* First feature whose ID equals block_id (the reference's attribute filter +
 * first feature, src/cn.c:162-171); returns its index or -1. */
int gcn10_blocks_find(const gcn10_blocks *b, int block_id);

/* ------------------------------------------------------------------------ */
/* rasters (src/raster.c:106-227) without GDAL                               */
/* ------------------------------------------------------------------------ */

typedef struct gcn10_raster gcn10_raster;   /* an open GeoTIFF or VRT, 1 band, Byte */

/* GeoTIFF (classic or BigTIFF; strips or tiles; none / LZW / DEFLATE /
 * PackBits; predictor 1 or 2) or a VRT mosaic of such files. */
gcn10_raster *gcn10_raster_open(const char *path, const char *vrt_tile_dir,
                                char *err, size_t errcap);
void gcn10_raster_close(gcn10_raster *r);
void gcn10_raster_info(const gcn10_raster *r, int *xsize, int *ysize, double gt[6]);
/* Rows [yoff, yoff+ycount) x columns [xoff, xoff+xcount) into dst (row-major,
 * xcount bytes per row).  Thread safe per raster handle.  0 or -1. */
int gcn10_raster_read(gcn10_raster *r, int xoff, int yoff, int xcount, int ycount,
                      uint8_t *dst, char *err, size_t errcap);

/* Georeferencing tags an output inherits from the landcover input (the
 * reference copies the input's WKT, src/raster.c:212-214). */
typedef struct gcn10_georef {
    uint16_t *geokeys;      /* GeoKeyDirectoryTag (34735) */
    int n_geokeys;
    double *geodoubles;     /* GeoDoubleParamsTag (34736) */
    int n_geodoubles;
    char *geoascii;         /* GeoAsciiParamsTag (34737), NUL terminated */
} gcn10_georef;
const gcn10_georef *gcn10_raster_georef(const gcn10_raster *r);

/* Streaming writer of one tiled DEFLATE GeoTIFF (GTiff, COMPRESS=DEFLATE,
 * TILED=YES -> 256x256 tiles, Byte, 1 band; src/raster.c:204-209).  Tiles may
 * be handed over already compressed (zlib streams), in any order. */
typedef struct gcn10_tiff_writer gcn10_tiff_writer;
gcn10_tiff_writer *gcn10_tiff_create(const char *path, int xsize, int ysize,
                                     const double gt[6], const gcn10_georef *georef,
                                     char *err, size_t errcap);
int gcn10_tiff_tiles_across(const gcn10_tiff_writer *w);
int gcn10_tiff_tiles_down(const gcn10_tiff_writer *w);
/* Appends one compressed tile (tx, ty) of `nbytes` zlib-stream bytes. */
int gcn10_tiff_put_tile(gcn10_tiff_writer *w, int tx, int ty, const void *zdata, size_t nbytes);
/* n tiles of the raster in one go (gathered writes); same result as n gcn10_tiff_put_tile calls */
int gcn10_tiff_put_tiles(gcn10_tiff_writer *w, int n, const int *tx, const int *ty,
                         const void *const *zdata, const uint32_t *nbytes);
/* n tiles whose streams lie in one extent of memory (stream i at data + rel_off[i]): one write for all of
 * them.  What the GPU encoders produce for a raster and a strip. */
int gcn10_tiff_put_extent(gcn10_tiff_writer *w, const void *data, size_t extent_bytes, int n, const int *tx,
                          const int *ty, const uint32_t *rel_off, const uint32_t *nbytes);
/* O_DIRECT for the tile data (config key "direct_io"): extents must then be 4096-aligned in memory and
 * readable to the next multiple of 4096.  0 = on, -1 = the file system refuses (nothing changed). */
int gcn10_tiff_set_direct(gcn10_tiff_writer *w, bool on);
/* Writes the directory and closes the file.  0 or -1. */
int gcn10_tiff_finish(gcn10_tiff_writer *w, char *err, size_t errcap);
void gcn10_tiff_abort(gcn10_tiff_writer *w);

/* zlib-compresses one 256x256 tile cut from a raster strip (rows are `stride`
 * bytes apart; the part of the tile outside the raster is zero-filled as GDAL
 * pads edge tiles).  Returns the compressed size or 0 on error. */
size_t gcn10_deflate_tile(const uint8_t *src, size_t stride, int valid_w, int valid_h,
                          int level, uint8_t *dst, size_t dstcap);

/* Convenience: whole raster in memory -> file (the reference's save_raster
 * signature, src/raster.c:192-194).  0 or -1. */
int gcn10_save_raster(const uint8_t *data, int xsize, int ysize, const double gt[6],
                      const gcn10_georef *georef, const char *path, int level,
                      char *err, size_t errcap);


/* ------------------------------------------------------------------------ */
/* the run: src/main.c:58-203 + process_block, src/cn.c:134-384              */
/* ------------------------------------------------------------------------ */

typedef struct gcn10_run_options {
    const char *config_path;    /* -c / --config (required)                      */
    const char *blocks_file;    /* -l / -b / --blocks, NULL = every shapefile ID */
    bool overwrite;             /* -o / --overwrite                              */
    int gpus;                   /* --gpus N, 0 = config key "gpus" or all visible */
    const char *lookups;        /* --lookups g_ii[,..]: overrides the config key "lookups"       */
    const char *conditions;     /* --conditions drained|undrained|both: overrides "conditions" */
} gcn10_run_options;

/* Runs the whole job: config, logs, block ids, lookup tables, one worker thread
 * per GPU pulling block ids from a shared atomic counter (replaces the static
 * round-robin over MPI ranks, src/main.c:171), per block: windows, index maps,
 * pinned-host strips double-buffered against the fused kernel, 18 tiled DEFLATE
 * GeoTIFFs named as src/cn.c:308, 341.  Returns the process exit code: 0, or 1
 * where the reference calls MPI_Abort(.., 1). */
int gcn10_run(const gcn10_run_options *opt);

/* Path of the HIP library this process would load (diagnostics). */
const char *gcn10_gpu_library_path(void);

#ifdef __cplusplus
}
#endif
#endif /* GCN10_HOST_H */

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

#ifdef __cplusplus
}
#endif
#endif /* GCN10_HOST_H */

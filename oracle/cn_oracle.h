/*
 * cn_oracle.h -- CPU oracle for the gcn10 curve-number hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED (see oracle/README.md):  the reference holds no
 * golden rasters, known-answer tests or checksums for this path, and its own
 * sources cannot be compiled in this image without stand-ins for gdal.h
 * (src/global.h:8-13), so this restatement is pinned only by (a) line-by-line
 * citation of the reference and (b) the shipped lookup CSVs, whose rows are
 * the per-(class, soil group) answers for unmodified soil codes.
 *
 * Every function cites the /root/reference file:line it restates.
 */
#ifndef CN_ORACLE_H
#define CN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/cn.c:13-85  load_lookup_table().  `path` is the full CSV file name
 * (the reference builds "<lookup_table_path>/default_lookup_<hc>_<arc>.csv",
 * src/cn.c:21).  Returns 0, or -1 if the file cannot be opened, -2 if it is
 * empty (both MPI_Abort in the reference, src/cn.c:28-33, 43-48).
 * *n_bad counts rows the reference would log as ERROR and skip. */
int oracle_load_lookup_table(const char *path, int table[256][5], int *n_bad);

/* src/cn.c:88-111  modify_hysogs_data(); drained != 0 <=> cond == "drained" */
void oracle_modify_hysogs_data(uint8_t *h, int npix, int drained);

/* src/cn.c:114-131 calculate_cn(); `out` must be pre-filled by the caller
 * (the reference memsets it to 255, src/cn.c:289). */
void oracle_calculate_cn(const uint8_t *esa, const uint8_t *hsg, int npix,
                         int table[256][5], uint8_t *out);

/* src/cn.c:218-232  nearest-neighbour HSG upsample onto the ESA grid. */
void oracle_resample(const uint8_t *coarse, int hsx, int hsy,
                     const double gt[6], const double soil_gt[6],
                     int esax, int esay, uint8_t *out);

/* The separable halves of the same loop: ci depends on x only, cj on y only
 * (src/cn.c:219-229).  Used to check the host-side index-map builder. */
void oracle_index_maps(const double gt[6], const double soil_gt[6],
                       int esax, int esay, int hsx, int hsy,
                       int32_t *ci, int32_t *cj);

/* src/raster.c:126-162  load_raster() window arithmetic for a raster with
 * geotransform t[6] and size rx x ry, clipped to bbox {minx,miny,maxx,maxy}.
 * Returns 0 and fills xoff,yoff,xcount,ycount,gt[6]; returns -1 for the
 * "invalid raster bounds" case (src/raster.c:142-147). */
int oracle_window(const double t[6], int rx, int ry, const double bbox[4],
                  int *xoff, int *yoff, int *xcount, int *ycount,
                  double gt[6]);

/* src/cn.c:205-380 in memory, I/O removed: resample once, then for
 * cond in {drained, undrained} x hc in {p,f,g} x arc in {i,ii,iii}
 * (src/cn.c:145-147,236-259) malloc+memcpy+modify+malloc+memset+lookup, with
 * the same per-raster allocations as the reference (src/cn.c:264,278).
 * tables[k] is the table for k = hi*3+ai.  out18[c*9+k] receives raster
 * (c,k); any out18 entry may be NULL (raster computed, then dropped), which is
 * what the cpu_baseline timing uses.  Returns 0 or -1 on allocation failure. */
int oracle_process_block_mem(const uint8_t *esa, int esax, int esay,
                             const double gt[6],
                             const uint8_t *coarse, int hsx, int hsy,
                             const double soil_gt[6],
                             int tables[9][256][5],
                             uint8_t *const out18[18]);

/* Same as above but only for the listed rasters: cond_mask bit0 = drained,
 * bit1 = undrained; table_mask bit k = table k.  Used to time subsets. */
int oracle_process_block_subset(const uint8_t *esa, int esax, int esay,
                                const double gt[6],
                                const uint8_t *coarse, int hsx, int hsy,
                                const double soil_gt[6],
                                int tables[9][256][5],
                                unsigned cond_mask, unsigned table_mask,
                                uint8_t *const out18[18]);

/* (int)double as the reference's x86-64 build performs it (cvttsd2si):
 * out-of-range and NaN give INT_MIN.  src/cn.c:225-226. */
int oracle_double_to_int_x86(double v);

#ifdef __cplusplus
}
#endif
#endif

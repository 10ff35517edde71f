/* host_internal.h -- declarations shared by the host C files, not part of the ABI. */
#ifndef GCN10_HOST_INTERNAL_H
#define GCN10_HOST_INTERNAL_H

#include "gcn10_host.h"

struct gcn10_tiff;      /* tiff.c: one open TIFF file */
struct gcn10_tiff *gcn10_tiff_open_reader(const char *path, char *err, size_t errcap);
void gcn10_tiff_close_reader(struct gcn10_tiff *t);
void gcn10_tiff_reader_info(const struct gcn10_tiff *t, int *xsize, int *ysize, double gt[6]);
const gcn10_georef *gcn10_tiff_reader_georef(const struct gcn10_tiff *t);
int gcn10_tiff_read_window(struct gcn10_tiff *t, int xoff, int yoff, int xcount, int ycount,
                           uint8_t *dst, size_t dst_stride, char *err, size_t errcap);

#endif

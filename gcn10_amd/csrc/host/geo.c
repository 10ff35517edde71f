/* geo.c -- the fp64 geotransform arithmetic of the curve-number generator.
 *
 * Everything here must reproduce the reference's doubles bit for bit, so this
 * file is compiled with -ffp-contract=off and keeps the reference's operation
 * order (/root/reference/src/cn.c:218-229, src/raster.c:126-162; the reference
 * is built -O3 without -march, src/CMakeLists.txt:68, i.e. SSE2 doubles, no
 * FMA).  The GPU never evaluates these expressions: hipcc would contract them.
 */
#include "gcn10_host.h"

#include <limits.h>
#include <math.h>

/* (int)double as the reference's x86-64 build does it (cvttsd2si): NaN and
 * values outside int give INT_MIN.  Written out so the result does not depend
 * on this host's handling of the undefined C conversion. */
#if defined(__GNUC__) && !defined(__clang__)
#define NO_CONTRACT __attribute__((optimize("fp-contract=off")))
#else
#define NO_CONTRACT
#pragma STDC FP_CONTRACT OFF
#endif

static int to_int_like_x86(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT_MIN;
    return (int)v;
}

static int clamp_to(int v, int n)
{
    if (v < 0)
        return 0;
    return v >= n ? n - 1 : v;
}

NO_CONTRACT void gcn10_build_index_maps(const double gt[6], const double soil_gt[6],
                            int W, int H, int hsx, int hsy,
                            int32_t *ci, int32_t *cj)
{
    const double sx0 = soil_gt[0], sdx = soil_gt[1];
    const double sy0 = soil_gt[3], sdy = fabs(soil_gt[5]);

    /* column map: src/cn.c:222-223, 225, 228 */
    for (int x = 0; x < W; x++) {
        double px = gt[0] + (x + 0.5) * gt[1];
        double dc = (px - sx0) / sdx;

        ci[x] = clamp_to(to_int_like_x86(round(dc)), hsx);
    }
    /* row map: src/cn.c:219, 224, 226, 229 */
    for (int y = 0; y < H; y++) {
        double py = gt[3] + (y + 0.5) * gt[5];
        double dr = (sy0 - py) / sdy;

        cj[y] = clamp_to(to_int_like_x86(round(dr)), hsy);
    }
}

NO_CONTRACT int gcn10_raster_window(const double t[6], int rx, int ry, const double bbox[4],
                        int *xoff, int *yoff, int *xcount, int *ycount,
                        double gt[6])
{
    /* bbox = {minx, miny, maxx, maxy}; src/raster.c:127-130 */
    int x0 = to_int_like_x86(floor((bbox[0] - t[0]) / t[1]));
    int y0 = to_int_like_x86(floor((bbox[3] - t[3]) / t[5]));
    int nx = to_int_like_x86(ceil((bbox[2] - bbox[0]) / t[1]));
    int ny = to_int_like_x86(ceil((bbox[1] - bbox[3]) / t[5]));

    if (x0 < 0) {           /* src/raster.c:134-141 */
        nx += x0;
        x0 = 0;
    }
    if (y0 < 0) {
        ny += y0;
        y0 = 0;
    }
    if (x0 >= rx || y0 >= ry || nx <= 0 || ny <= 0)
        return -1;          /* "invalid raster bounds", src/raster.c:142-147 */
    if (x0 + nx > rx)       /* src/raster.c:148-153 */
        nx = rx - x0;
    if (y0 + ny > ry)
        ny = ry - y0;

    *xoff = x0;
    *yoff = y0;
    *xcount = nx;
    *ycount = ny;
    gt[0] = t[0] + x0 * t[1];       /* src/raster.c:157-162 */
    gt[1] = t[1];
    gt[2] = t[2];
    gt[3] = t[3] + y0 * t[5];
    gt[4] = t[4];
    gt[5] = t[5];
    return 0;
}

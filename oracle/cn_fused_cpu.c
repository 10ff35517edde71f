/* cn_fused_cpu.c -- the "best CPU" line of bench.py's cpu_baseline.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see cn_oracle.h); PARITY UNPINNED like the rest of
 * oracle/.  One fused pass per block instead of the reference's per-raster
 * malloc + memcpy + modify + memset + lookup (src/cn.c:236-290): the separable index maps of
 * src/cn.c:218-229 (computed by oracle_index_maps, never here: this file is built with
 * -march=native and must not evaluate the fp64 expressions), the soil remap of src/cn.c:92-110 as
 * two 256-entry maps, and byte tables with the `< 255` rule of src/cn.c:125-128 folded in.
 * tests/test_oracle.py checks it against oracle_process_block_subset byte for byte.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_fused_block(const uint8_t *esa, int W, int H, const uint8_t *coarse, int hsx,
                       const int32_t *ci, const int32_t *cj, int tables[9][256][5],
                       unsigned cond_mask, unsigned table_mask, uint8_t *const out18[18])
{
    /* T8[k][s][lc], s = 0..4 soil group, 5 = "not a soil group" -> 255 (src/cn.c:123-124) */
    static _Thread_local uint8_t T8[9][6][256];
    uint8_t plane[2][256];
    uint8_t *hrow[2];
    int last_r = -1;

    memset(T8, 255, sizeof T8);
    for (int k = 0; k < 9; k++)
        for (int lc = 0; lc < 256; lc++)
            for (int s = 0; s < 5; s++) {
                int v = tables[k][lc][s];
                T8[k][s][lc] = v < 255 ? (uint8_t)v : 255;      /* src/cn.c:125-128, :289 */
            }
    for (int h = 0; h < 256; h++) {
        int dual = h >= 11 && h <= 14;
        int d = dual ? 4 : h, u = dual ? h - 10 : h;            /* src/cn.c:92-110 */
        plane[0][h] = (uint8_t)(d < 5 ? d : 5);
        plane[1][h] = (uint8_t)(u < 5 ? u : 5);
    }
    hrow[0] = malloc((size_t)W ? (size_t)W : 1);
    hrow[1] = malloc((size_t)W ? (size_t)W : 1);
    if (!hrow[0] || !hrow[1]) {
        free(hrow[0]);
        free(hrow[1]);
        return -1;
    }
    for (int y = 0; y < H; y++) {
        const uint8_t *e = esa + (size_t)y * W;
        if (cj[y] != last_r) {          /* a coarse row serves ~25 raster rows */
            const uint8_t *crow = coarse + (size_t)cj[y] * hsx;
            for (int x = 0; x < W; x++) {
                uint8_t h = crow[ci[x]];                        /* src/cn.c:230 */
                hrow[0][x] = plane[0][h];
                hrow[1][x] = plane[1][h];
            }
            last_r = cj[y];
        }
        for (int c = 0; c < 2; c++) {
            if (!(cond_mask & (1u << c)))
                continue;
            const uint8_t *hs = hrow[c];
            for (int k = 0; k < 9; k++) {
                uint8_t *o = out18[c * 9 + k];
                if (!(table_mask & (1u << k)) || !o)
                    continue;
                o += (size_t)y * W;
                const uint8_t *t = &T8[k][0][0];
                for (int x = 0; x < W; x++)
                    o[x] = t[(unsigned)hs[x] * 256u + e[x]];
            }
        }
    }
    free(hrow[0]);
    free(hrow[1]);
    return 0;
}

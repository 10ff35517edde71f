/*
 * cn_oracle.c -- CPU restatement of gcn10's curve-number hot path.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see cn_oracle.h.
 *
 * Build like the reference builds its own sources (src/CMakeLists.txt:68,
 * "-Wall -O3", C99, no -march, hence no FMA contraction on x86-64);
 * oracle/Makefile adds -ffp-contract=off so the fp64 index arithmetic cannot
 * be fused on any host.
 */
#define _GNU_SOURCE
#include "cn_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NODATA 255

/* keeps a raster that nobody reads (timing runs pass NULL outputs) from being
 * optimised away together with its malloc/free pair */
#define KEEP_ALIVE(p) __asm__ volatile("" : : "r"(p) : "memory")

/* ------------------------------------------------------------------ */
/* src/cn.c:13-85                                                      */
int oracle_load_lookup_table(const char *path, int table[256][5], int *n_bad)
{
    /* 128-byte line buffer as in src/cn.c:17: fgets() cuts longer lines
     * into several "rows", each parsed on its own. */
    char row[128];
    int bad = 0;
    FILE *fp = fopen(path, "r");

    if (n_bad)
        *n_bad = 0;
    if (!fp)
        return -1;                      /* src/cn.c:28-33 */

    for (int lc = 0; lc < 256; lc++)    /* src/cn.c:36-40 */
        for (int sg = 0; sg < 5; sg++)
            table[lc][sg] = NODATA;

    if (!fgets(row, sizeof row, fp)) {  /* header, src/cn.c:43-48 */
        fclose(fp);
        return -2;
    }

    while (fgets(row, sizeof row, fp)) {        /* src/cn.c:51 */
        char *save = NULL;
        char *code = strtok_r(row, ",", &save); /* src/cn.c:52 */
        char *sep, *val;
        int lc, sg, cn;

        if (!code)
            continue;                           /* src/cn.c:53-55 */
        sep = strchr(code, '_');                /* src/cn.c:57 */
        if (!sep) {                             /* src/cn.c:58-63 */
            bad++;
            continue;
        }
        *sep = '\0';
        lc = atoi(code);                        /* src/cn.c:65 */
        switch (sep[1]) {                       /* src/cn.c:66 */
        case 'A': sg = 1; break;
        case 'B': sg = 2; break;
        case 'C': sg = 3; break;
        default:  sg = 4; break;    /* 'D' and everything else */
        }
        val = strtok_r(NULL, ",", &save);       /* src/cn.c:67 */
        if (!val) {                             /* src/cn.c:68-73 */
            bad++;
            continue;
        }
        cn = atoi(val);                         /* src/cn.c:74 */
        if (lc >= 0 && lc < 256)                /* src/cn.c:75-77 */
            table[lc][sg] = cn;
        else
            bad++;                              /* src/cn.c:78-82 */
    }
    fclose(fp);
    if (n_bad)
        *n_bad = bad;
    return 0;
}

/* ------------------------------------------------------------------ */
/* src/cn.c:88-111                                                     */
void oracle_modify_hysogs_data(uint8_t *h, int npix, int drained)
{
    if (drained) {
        /* dual classes A/D..D/D all act as D, src/cn.c:92-98 */
        for (int i = 0; i < npix; i++)
            if (h[i] >= 11 && h[i] <= 14)
                h[i] = 4;
    }
    else {
        /* dual classes act as their first letter, src/cn.c:99-110 */
        for (int i = 0; i < npix; i++) {
            switch (h[i]) {
            case 11: h[i] = 1; break;
            case 12: h[i] = 2; break;
            case 13: h[i] = 3; break;
            case 14: h[i] = 4; break;
            default: break;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* src/cn.c:114-131                                                    */
void oracle_calculate_cn(const uint8_t *esa, const uint8_t *hsg, int npix,
                         int table[256][5], uint8_t *out)
{
    for (int i = 0; i < npix; i++) {
        int lc = esa[i];
        int sg = hsg[i];

        /* lc is a uint8 so 0 <= lc < 256 always holds (src/cn.c:123) */
        if (sg < 5) {
            int v = table[lc][sg];              /* src/cn.c:125 */

            if (v < NODATA)                     /* src/cn.c:126 */
                out[i] = (uint8_t)v;            /* truncating cast, :127 */
        }
    }
}

/* ------------------------------------------------------------------ */
int oracle_double_to_int_x86(double v)
{
    /* cvttsd2si: the "integer indefinite" value for NaN / out of range */
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT_MIN;
    return (int)v;
}

static inline int clamp_index(int v, int n)
{
    /* src/cn.c:228-229 */
    return v < 0 ? 0 : (v >= n ? n - 1 : v);
}

/* src/cn.c:218-232                                                    */
void oracle_resample(const uint8_t *coarse, int hsx, int hsy,
                     const double gt[6], const double soil_gt[6],
                     int esax, int esay, uint8_t *out)
{
    for (int y = 0; y < esay; y++) {
        double py = gt[3] + (y + 0.5) * gt[5];          /* :219 */

        for (int x = 0; x < esax; x++) {
            double px = gt[0] + (x + 0.5) * gt[1];      /* :222 */
            double dc = (px - soil_gt[0]) / soil_gt[1]; /* :223 */
            double dr = (soil_gt[3] - py) / fabs(soil_gt[5]);   /* :224 */
            int ci = clamp_index(oracle_double_to_int_x86(round(dc)), hsx);
            int cj = clamp_index(oracle_double_to_int_x86(round(dr)), hsy);

            /* int index arithmetic as in the reference, :230 */
            out[y * esax + x] = coarse[cj * hsx + ci];
        }
    }
}

void oracle_index_maps(const double gt[6], const double soil_gt[6],
                       int esax, int esay, int hsx, int hsy,
                       int32_t *ci, int32_t *cj)
{
    for (int x = 0; x < esax; x++) {
        double px = gt[0] + (x + 0.5) * gt[1];
        double dc = (px - soil_gt[0]) / soil_gt[1];

        ci[x] = clamp_index(oracle_double_to_int_x86(round(dc)), hsx);
    }
    for (int y = 0; y < esay; y++) {
        double py = gt[3] + (y + 0.5) * gt[5];
        double dr = (soil_gt[3] - py) / fabs(soil_gt[5]);

        cj[y] = clamp_index(oracle_double_to_int_x86(round(dr)), hsy);
    }
}

/* ------------------------------------------------------------------ */
/* src/raster.c:126-162                                                */
int oracle_window(const double t[6], int rx, int ry, const double bbox[4],
                  int *xoff, int *yoff, int *xcount, int *ycount,
                  double gt[6])
{
    int xo = oracle_double_to_int_x86(floor((bbox[0] - t[0]) / t[1]));  /* :127 */
    int yo = oracle_double_to_int_x86(floor((bbox[3] - t[3]) / t[5]));  /* :128 */
    int xc = oracle_double_to_int_x86(ceil((bbox[2] - bbox[0]) / t[1])); /* :129 */
    int yc = oracle_double_to_int_x86(ceil((bbox[1] - bbox[3]) / t[5])); /* :130 */

    if (xo < 0) {               /* :134-137 */
        xc += xo;
        xo = 0;
    }
    if (yo < 0) {               /* :138-141 */
        yc += yo;
        yo = 0;
    }
    if (xo >= rx || yo >= ry || xc <= 0 || yc <= 0)     /* :142-147 */
        return -1;
    if (xo + xc > rx)           /* :148-150 */
        xc = rx - xo;
    if (yo + yc > ry)           /* :151-153 */
        yc = ry - yo;

    *xoff = xo;
    *yoff = yo;
    *xcount = xc;
    *ycount = yc;
    gt[0] = t[0] + xo * t[1];   /* :157-162 */
    gt[1] = t[1];
    gt[2] = t[2];
    gt[3] = t[3] + yo * t[5];
    gt[4] = t[4];
    gt[5] = t[5];
    return 0;
}

/* ------------------------------------------------------------------ */
/* src/cn.c:205-380 with the I/O taken out                             */
int oracle_process_block_subset(const uint8_t *esa, int esax, int esay,
                                const double gt[6],
                                const uint8_t *coarse, int hsx, int hsy,
                                const double soil_gt[6],
                                int tables[9][256][5],
                                unsigned cond_mask, unsigned table_mask,
                                uint8_t *const out18[18])
{
    int npix = esax * esay;                     /* int, as src/cn.c:208 */
    uint8_t *fine = malloc((size_t)npix);       /* src/cn.c:209 */

    if (!fine)
        return -1;
    oracle_resample(coarse, hsx, hsy, gt, soil_gt, esax, esay, fine);

    for (int c = 0; c < 2; c++) {               /* drained, undrained :145,236 */
        if (!(cond_mask & (1u << c)))
            continue;
        for (int k = 0; k < 9; k++) {           /* hc-major, arc-minor :258-259 */
            uint8_t *adj, *cn;

            if (!(table_mask & (1u << k)))
                continue;
            adj = malloc((size_t)npix);         /* src/cn.c:264 */
            if (!adj) {
                free(fine);
                return -1;
            }
            memcpy(adj, fine, (size_t)npix);    /* src/cn.c:274 */
            oracle_modify_hysogs_data(adj, npix, c == 0);       /* :275 */

            cn = malloc((size_t)npix);          /* src/cn.c:278 */
            if (!cn) {
                free(adj);
                free(fine);
                return -1;
            }
            memset(cn, NODATA, (size_t)npix);   /* src/cn.c:289 */
            oracle_calculate_cn(esa, adj, npix, tables[k], cn); /* :290 */

            KEEP_ALIVE(cn);
            if (out18 && out18[c * 9 + k])
                memcpy(out18[c * 9 + k], cn, (size_t)npix);
            free(cn);                           /* src/cn.c:376-377 */
            free(adj);
        }
    }
    free(fine);                                 /* src/cn.c:383 */
    return 0;
}

int oracle_process_block_mem(const uint8_t *esa, int esax, int esay,
                             const double gt[6],
                             const uint8_t *coarse, int hsx, int hsy,
                             const double soil_gt[6],
                             int tables[9][256][5],
                             uint8_t *const out18[18])
{
    return oracle_process_block_subset(esa, esax, esay, gt, coarse, hsx, hsy,
                                       soil_gt, tables, 3u, 0x1ffu, out18);
}

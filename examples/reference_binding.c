/* reference_binding.c -- the binding INTEGRATION.md section 2 sketches, as a file that compiles.
 *
 * What a maintainer of clawrim/gcn10 would put in place of src/cn.c:205-290 (the resample
 * loop and the 18 memcpy + modify_hysogs_data + memset + calculate_cn passes), keeping
 * load_raster() / save_raster() and everything else of the reference as it is:
 *
 *     gcn10_binding_setup(rank, lookup_dir)        once, after MPI_Init
 *     gcn10_binding_block(esa, ..., cn_out)        per block, between load_raster and save_raster
 *
 * Plain C against include/gcn10_gpu.h and include/gcn10_host.h; no GDAL, no MPI in here.
 * With -DGCN10_BINDING_MAIN it is also a small program (tests/test_binding_example.py):
 *     reference_binding <lookup_dir> <W> <H> <hsx> <hsy> <in.bin> <out.bin>
 * in.bin = 6 doubles gt, 6 doubles soil_gt, W*H landcover bytes, hsx*hsy soil bytes;
 * out.bin = the 18 rasters, cond-major like src/cn.c:236-259.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gcn10_gpu.h"
#include "gcn10_host.h"

static gcn10_gpu_ctx *ctx;              /* one per MPI rank = one per GPU */
static int tables[9][256][5];

static void on_bad_row(void *user, const char *message)
{
    (void)user;
    fprintf(stderr, "%s\n", message);    /* the reference: log_message("ERROR", ..., true) */
}

/* once per rank.  0, or -1 where the reference calls MPI_Abort (src/cn.c:25, 32) */
int gcn10_binding_setup(int rank, const char *lookup_dir)
{
    int n_dev, failed = -1;

    /* the library on the library path must be the one this file was compiled against (ABI 3: tile flags,
     * one extent per raster and strip; include/gcn10_gpu.h) */
    if (gcn10_gpu_abi_version() != GCN10_GPU_ABI_VERSION) {
        fprintf(stderr, "libgcn10_gpu.so has ABI version %d, this binding was built for %d\n", gcn10_gpu_abi_version(),
                GCN10_GPU_ABI_VERSION);
        return -1;
    }
    n_dev = gcn10_gpu_device_count();
    if (n_dev <= 0 || gcn10_gpu_init(rank % n_dev, &ctx) != 0) {
        fprintf(stderr, "gpu: %s\n", gcn10_gpu_last_error());     /* no CPU fallback */
        return -1;
    }
    /* the 9 CSVs once, not 18 times per block (src/cn.c:261) */
    if (gcn10_load_all_lookup_tables(lookup_dir, tables, &failed, on_bad_row, NULL) != 0) {
        fprintf(stderr, "lookup table %d of %s cannot be loaded\n", failed, lookup_dir);
        return -1;
    }
    return gcn10_gpu_set_tables(ctx, &tables[0][0][0], 9) == 0 ? 0 : -1;
}

/* One block: esa[esay][esax] and its geotransform from load_raster(), the coarse soil window
 * hysogs[hsy][hsx] and its geotransform likewise; cn_out[r] (r = cond * 9 + hc * 3 + arc) are 18
 * caller-owned rasters of esax * esay bytes, ready for save_raster().  0 or -1. */
int gcn10_binding_block(const uint8_t *esa, int esax, int esay, const double gt[6], const uint8_t *hysogs,
                        int hsx, int hsy, const double soil_gt[6], uint8_t *const cn_out[GCN10_N_RASTERS])
{
    size_t npix = (size_t)esax * (size_t)esay;
    int32_t *ci = malloc((size_t)esax * sizeof *ci), *cj = malloc((size_t)esay * sizeof *cj);
    uint8_t *d_esa = NULL, *d_coarse = NULL, *d_out[GCN10_N_RASTERS] = { 0 };
    int32_t *d_ci = NULL, *d_cj = NULL;
    int rc = -1;

    if (!ci || !cj)
        goto out;
    gcn10_build_index_maps(gt, soil_gt, esax, esay, hsx, hsy, ci, cj);          /* src/cn.c:219-229 */
    if (gcn10_gpu_malloc(ctx, npix, (void **)&d_esa) != 0 ||
        gcn10_gpu_malloc(ctx, (size_t)hsx * hsy, (void **)&d_coarse) != 0 ||
        gcn10_gpu_malloc(ctx, (size_t)esax * 4, (void **)&d_ci) != 0 ||
        gcn10_gpu_malloc(ctx, (size_t)esay * 4, (void **)&d_cj) != 0)
        goto out;
    for (int r = 0; r < GCN10_N_RASTERS; r++)
        if (gcn10_gpu_malloc(ctx, npix, (void **)&d_out[r]) != 0)
            goto out;
    if (gcn10_gpu_memcpy_h2d(ctx, d_esa, esa, npix, NULL) != 0 ||
        gcn10_gpu_memcpy_h2d(ctx, d_coarse, hysogs, (size_t)hsx * hsy, NULL) != 0 ||
        gcn10_gpu_memcpy_h2d(ctx, d_ci, ci, (size_t)esax * 4, NULL) != 0 ||
        gcn10_gpu_memcpy_h2d(ctx, d_cj, cj, (size_t)esay * 4, NULL) != 0 ||
        gcn10_gpu_prepare_tile(ctx, d_coarse, hsx, hsy, d_ci, esax, NULL) != 0 ||
        gcn10_gpu_cn_strip(ctx, d_esa, esax, esay, d_cj, GCN10_COND_DRAINED | GCN10_COND_UNDRAINED, 0x1ff,
                           d_out, NULL) != 0)
        goto out;
    for (int r = 0; r < GCN10_N_RASTERS; r++)       /* same order as src/cn.c:236-259 */
        if (gcn10_gpu_memcpy_d2h(ctx, cn_out[r], d_out[r], npix, NULL) != 0)
            goto out;
    rc = gcn10_gpu_stream_sync(ctx, NULL) == 0 ? 0 : -1;
out:
    if (rc != 0)
        fprintf(stderr, "gpu: %s\n", gcn10_gpu_last_error());
    for (int r = 0; r < GCN10_N_RASTERS; r++)
        if (d_out[r]) gcn10_gpu_free(ctx, d_out[r]);
    if (d_esa) gcn10_gpu_free(ctx, d_esa);
    if (d_coarse) gcn10_gpu_free(ctx, d_coarse);
    if (d_ci) gcn10_gpu_free(ctx, d_ci);
    if (d_cj) gcn10_gpu_free(ctx, d_cj);
    free(ci);
    free(cj);
    return rc;
}

void gcn10_binding_teardown(void)
{
    if (ctx)
        gcn10_gpu_destroy(ctx);
    ctx = NULL;
}

#ifdef GCN10_BINDING_MAIN
int main(int argc, char **argv)
{
    double gt[6], sgt[6];
    uint8_t *esa, *soil, *out[GCN10_N_RASTERS];
    FILE *f;
    int W, H, hsx, hsy, rc = 1;

    if (argc != 8)
        return 2;
    W = atoi(argv[2]);
    H = atoi(argv[3]);
    hsx = atoi(argv[4]);
    hsy = atoi(argv[5]);
    esa = malloc((size_t)W * H);
    soil = malloc((size_t)hsx * hsy);
    f = fopen(argv[6], "rb");
    if (!f || !esa || !soil || fread(gt, 8, 6, f) != 6 || fread(sgt, 8, 6, f) != 6 ||
        fread(esa, 1, (size_t)W * H, f) != (size_t)W * H || fread(soil, 1, (size_t)hsx * hsy, f) != (size_t)hsx * hsy)
        return 2;
    fclose(f);
    for (int r = 0; r < GCN10_N_RASTERS; r++)
        out[r] = malloc((size_t)W * H);
    if (gcn10_binding_setup(0, argv[1]) == 0 && gcn10_binding_block(esa, W, H, gt, soil, hsx, hsy, sgt, out) == 0) {
        f = fopen(argv[7], "wb");
        for (int r = 0; f && r < GCN10_N_RASTERS; r++)
            fwrite(out[r], 1, (size_t)W * H, f);
        if (f && fclose(f) == 0)
            rc = 0;
    }
    gcn10_binding_teardown();
    return rc;
}
#endif

/* gpuapi.c -- binds include/gcn10_gpu.h at run time (dlopen).
 *
 * The host library must load on machines without ROCm (config / I-O tests,
 * `gcn10 --help`), so the HIP library is opened on first use.  There is no CPU
 * fallback: when it cannot be loaded the run stops with an error.
 */
#include "host_internal.h"

#include <dlfcn.h>
#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static struct gcn10_gpu_api g_api;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static char g_path[PATH_MAX] = "";
static char g_err[PATH_MAX + 512] = "";

const char *gcn10_gpu_library_path(void)
{
    if (!g_path[0]) {
        const char *env = getenv("GCN10_GPU_LIB");
        Dl_info info;

        if (env && *env) {
            snprintf(g_path, sizeof g_path, "%s", env);
        }
        else if (dladdr((void *)gcn10_gpu_library_path, &info) && info.dli_fname) {
            /* next to libgcn10_host.so (or the gcn10 binary's ../gcn10_amd) */
            const char *slash = strrchr(info.dli_fname, '/');
            int n = slash ? (int)(slash - info.dli_fname) : 0;

            if (n > 0)
                snprintf(g_path, sizeof g_path, "%.*s/libgcn10_gpu.so", n, info.dli_fname);
            else
                snprintf(g_path, sizeof g_path, "libgcn10_gpu.so");
        }
        else {
            snprintf(g_path, sizeof g_path, "libgcn10_gpu.so");
        }
    }
    return g_path;
}

static void load_once(void)
{
    void *h;

    /* Hardware queues of this process on a device: the ROCm runtime maps its streams onto 4 by default, and streams
     * that share one run in order.  With two workers per GPU the program has six busy streams (kernels, copy-back
     * and input of each worker): on 4 queues the two workers' kernel streams landed on the same one
     * (profiles/r03/pipeline_kernel_overlap_*: "by (queue, stream)").  16 queues, 72 blocks, steady state: noisy
     * blocks 0.0183 -> 0.0164 s, patchy ones with files 0.0089 -> 0.0080 (profiles/r03/hw_queues_72_blocks.txt).
     * Set before the HIP runtime is loaded; a value the user has set stays. */
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    h = dlopen(gcn10_gpu_library_path(), RTLD_NOW | RTLD_LOCAL);

    if (!h)
        h = dlopen("libgcn10_gpu.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g_err, sizeof g_err, "cannot load %s: %s", g_path, dlerror());
        return;
    }
#define BIND(name)                                                          \
    do {                                                                    \
        *(void **)(&g_api.name) = dlsym(h, "gcn10_gpu_" #name);             \
        if (!g_api.name) {                                                  \
            snprintf(g_err, sizeof g_err, "%s lacks gcn10_gpu_" #name, g_path); \
            return;                                                         \
        }                                                                   \
    } while (0)
    BIND(abi_version); BIND(device_count); BIND(init); BIND(destroy); BIND(last_error);
    BIND(device_info); BIND(malloc); BIND(free); BIND(host_alloc); BIND(host_free);
    BIND(memcpy_h2d); BIND(memcpy_d2h); BIND(memset); BIND(stream_create); BIND(stream_destroy);
    BIND(stream_sync); BIND(device_sync); BIND(event_create); BIND(event_destroy);
    BIND(event_record); BIND(event_sync); BIND(stream_wait_event); BIND(event_elapsed_ms);
    BIND(set_tables); BIND(prepare_tile); BIND(cn_strip);
    BIND(deflate_arena_bound); BIND(deflate_strip); BIND(pci_bus_id);
    BIND(deflate_fused_strip); BIND(deflate_fused_available); BIND(inflate_tiles);
#undef BIND
    *(void **)(&g_api.set_option) = dlsym(h, "gcn10_gpu_set_option");    /* tuning only: may be absent (tests' stand-in) */
    if (g_api.abi_version() != GCN10_GPU_ABI_VERSION) {
        snprintf(g_err, sizeof g_err, "%s has ABI version %d, expected %d", g_path,
                 g_api.abi_version(), GCN10_GPU_ABI_VERSION);
        return;
    }
    g_api.loaded = true;
}

const struct gcn10_gpu_api *gcn10_gpu_api_get(char *err, size_t errcap)
{
    pthread_once(&g_once, load_once);
    if (!g_api.loaded) {
        snprintf(err, errcap, "%s (the CN path has no CPU fallback)", g_err);
        return NULL;
    }
    return &g_api;
}

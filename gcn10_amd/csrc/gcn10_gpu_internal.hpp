// gcn10_gpu_internal.hpp -- shared by the translation units of libgcn10_gpu.so.
#ifndef GCN10_GPU_INTERNAL_HPP
#define GCN10_GPU_INTERNAL_HPP

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gcn10_gpu.h"

namespace gcn10 {

// error plumbing: message of the calling thread (gcn10_gpu_last_error)
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return gcn10::fail(e_ == hipErrorOutOfMemory ? GCN10_E_NOMEM : GCN10_E_HIP, \
                               "%s: %s", #expr, hipGetErrorString(e_));                 \
    } while (0)

int use_device(gcn10_gpu_ctx *ctx);
hipStream_t as_stream(gcn10_gpu_ctx *ctx, gcn10_stream_t s);

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// soil code byte of the hx workspace: drained plane | undrained plane << 4, planes 0..5;
// compact index = drained * 6 + undrained
constexpr int kClassCodes = 36;

}  // namespace gcn10

struct gcn10_gpu_ctx {
    int device = -1;
    int n_cus = 0;
    hipStream_t main_stream = nullptr;
    uint8_t *d_lut16 = nullptr;     // kLut16Bytes
    uint8_t *d_lut1 = nullptr;      // 9 * kLut1Bytes
    int n_tables = 0;
    uint8_t *d_hx = nullptr;        // row 0 of the soil-code workspace (= d_hx_alloc + 16)
    uint8_t *d_hx_alloc = nullptr;
    uint32_t *d_hx4 = nullptr;          // compact soil words, one per 16-px column group and coarse row
    uint32_t *d_hx4_complex = nullptr;  // device word: == hx4_gen when some group of the prepared tile has no compact form
    uint32_t hx4_gen = 0;               // generation number of the prepared tile
    bool hx4_ready = false;             // the last gcn10_gpu_prepare_tile wrote the words
    int compact_soil = 1;               // option: write and use the compact words
    size_t hx_capacity = 0;
    uint32_t hx_stride = 0;
    uint32_t hx_W = 0;
    uint32_t hx_rows = 0;
    const char *last_kernel = "";
    char kernel_name[96] = "";
    // tuning knobs (gcn10_gpu_set_option); defaults = the round-1 measured best
    int grid_blocks_per_cu = 16;
    int ilp16 = 0;          // sub-chunks per loop trip, all-tables kernel (1, 2; 0 = by stream count)
    int ilp1 = 2;           // same, single-table kernel (1, 2, 4)
    int nontemporal = 1;
    int xcd_slabs = 1;
    int prefetch = 1;       // software pipeline: loads of the next trip issued before the current one is
                            // consumed (two register sets; -1 = default = on)
    // launch shape of the single-table strip kernel as gcn10_gpu_tune_single_raster left it (the knobs
    // above keep steering the all-tables kernel)
    struct SingleShape {
        bool set = false;
        int xcd_slabs = 1, grid_blocks_per_cu = 8, ilp = 2, prefetch = 1;
    } single;
    hipEvent_t time_start = nullptr, time_stop = nullptr;  // one-shot: bracket the next strip kernel
    uint8_t *d_class_of = nullptr;  // [36][256] pixel class of (soil code, landcover), then [18][256] values
    int n_classes = 0;              // 0: not available (set_tables not called, or > 256 classes)
    int arena_segment_align = 4096; // tile encoder: a raster's streams of a strip start at a multiple of this (16 .. 4096)
    int deflate_wave_codes = 1;     // pass B of the tile encoder: 1 = one wave per tile, 0 = one thread
    int fused_parse = 1;            // pass F-A of the fused encoder: 1 = one lane per 64-pixel segment (round 3), 0 = one lane per row
    int fused_emit = 1;             // pass F-C of the fused encoder: 1 = every wave packs its own quarter of the tokens (round 3), 0 = lock step
    int codes_stop = 0;             // timing experiments only: pass B leaves after phase (value - 1)
    int fused_stats_stop = 0;       // timing experiments only: pass F-A leaves after phase (value - 1)
    int fused_diag = 0;             // timing experiments only (streams become invalid): 2 = pass F-C
                                    // without its token trips (set-up cost alone)
    int event_sync_sleep_us = 0;    // gcn10_gpu_event_sync: 0 = hipEventSynchronize (spins); n > 0 = query, sleep n us, query ...
    int inflate_diag = 0;           // timing experiments only (output invalid): 1 = copier idle, 2 = empty batches
    bool fused_ready = false;
    bool codes_ready = false;       // LDS attribute of the per-thread code construction set
    bool deflate_ready = false;     // LDS attributes of the tile encoder set on this device
    void *deflate_ws = nullptr;     // per-tile statistics + code books of the tile encoder
    size_t deflate_ws_cap = 0;
    void *inflate_ws = nullptr;     // linear slots of the tiles being decoded (gcn10_inflate.hip)
    size_t inflate_ws_cap = 0;
};

#endif

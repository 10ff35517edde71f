// gcn10_gpu.hip -- MI355X (gfx950 / CDNA4) curve-number engine behind the C ABI
// of include/gcn10_gpu.h.  Written for gfx950 only: 64-lane waves, 16-byte
// per-lane global accesses (1 KiB per wave instruction), lookup tables in LDS.
//
// Reference behaviour restated here (citations are /root/reference paths):
//   src/cn.c:218-232   HSG nearest-neighbour upsample  -> x-expand + row pick
//   src/cn.c:88-111    modify_hysogs_data              -> soil-code byte (sD | sU<<4)
//   src/cn.c:114-131   calculate_cn                    -> LDS table gather
//   src/cn.c:236-290   18 (cond, hc, arc) passes       -> one pass, 18 store streams
//
// Data layout in HBM (all rasters uint8, row-major, exactly the reference's
// malloc'd buffers, src/raster.c:169-178):
//   esa      [rows*W]            landcover strip, read once, 16 B / lane
//   out[r]   [rows*W]  r < 18    CN rasters, written once, 16 B / lane, nontemporal
//   hx       [hsy][hx_stride]    soil window expanded along x only, one byte per
//                                fine column holding both remapped soil groups;
//                                52 MB for a 36000-wide block, re-read ~25x per
//                                row from L2 / Infinity Cache, once from HBM
//   lut16    [6][256] x 16 B     row (s, lc): bytes 0..8 = CN of tables 0..8,
//                                plane s=5 is all 255 (soil group not in 0..4)
//   lut1[k]  [6][256] x 1 B      the same for a single table k
//
// The dominant kernel (cn_strip_*) is HBM-bound: 1 + n_out bytes per pixel of
// DRAM traffic against ~20 VALU ops and 2 ds_read_b128 per pixel.

#include <hip/hip_runtime.h>
#include <unistd.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "gcn10_gpu.h"
#include "gcn10_gpu_internal.hpp"

using gcn10::as_stream;
using gcn10::fail;
using gcn10::u32x4;
using gcn10::use_device;

namespace {

// ------------------------------------------------------------------------
// constants shared by host and device
// ------------------------------------------------------------------------
constexpr int kThreads = 256;           // 4 waves per workgroup
constexpr int kPxPerLane = 16;          // one dwordx4 per lane per stream
constexpr int kChunk = kThreads * kPxPerLane;   // 4096 px per workgroup step
constexpr int kPlanes = 6;              // soil groups 0..4 + "invalid" plane
constexpr int kInvalidPlane = 5;
// LDS plane stride for the 16-byte-row table: 256 rows + one row of padding so
// that equal landcover classes in different planes fall in different banks.
constexpr int kPlane16 = 256 * 16 + 16;
constexpr int kLut16Bytes = kPlanes * kPlane16;
// LDS plane stride for the single-table byte LUT (+4 B pad: next bank).
constexpr int kPlane1 = 256 + 4;
constexpr int kLut1Bytes = kPlanes * kPlane1;


struct StripParams {
    const uint8_t *esa;
    const uint8_t *hx;      // x-expanded soil codes
    const int32_t *cj;      // coarse row of every strip row
    const uint8_t *lut;     // device image of the LDS table (lut16 or lut1[k])
    uint8_t *out[GCN10_N_RASTERS];
    uint32_t W;
    uint32_t rows;
    uint32_t npix;          // W*rows (< 2^31, as the reference's int npix, src/cn.c:208)
    uint32_t nvec16;        // npix rounded down to a multiple of 16: the part done in 16-byte lane groups
    uint32_t hx_stride;
    uint32_t hx_rows;       // rows in hx; cj is clamped to it defensively
    uint32_t nchunks;
    uint32_t table_mask;
    uint32_t single_k;      // table index for the single-table variant
    uint32_t xcd_slabs;     // 1: each XCD streams a contiguous eighth of the strip
    // compact soil words (one dword per 16-px column group, see expand_x_codes); hx4 = null: not in use
    const uint32_t *hx4;
    const uint32_t *hx4_complex;    // device word: == hx4_gen when some group of the tile has no compact form
    uint32_t hx4_stride;            // dwords per soil row
    uint32_t hx4_gen;               // generation number of the prepared tile (never 0)
};

// soil code byte: low nibble = plane for "drained", high nibble = "undrained".
// src/cn.c:92-110 followed by the `soil_group < 5` test of src/cn.c:123-124.
__host__ __device__ inline uint8_t soil_code(uint8_t h)
{
    uint8_t d, u;
    if (h >= 11 && h <= 14) {
        d = 4;                  // drained: every dual class acts as D
        u = (uint8_t)(h - 10);  // undrained: 11..14 -> 1..4
    }
    else {
        d = u = (h < 5) ? h : (uint8_t)kInvalidPlane;
    }
    return (uint8_t)(d | (u << 4));
}

// ------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------

// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {hi:lo}
// (selector 0..3 -> lo bytes 0..3, 4..7 -> hi bytes 0..3).
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// 4x4 byte transpose: in a,b,c,d = one dword (4 table values) of pixels 0..3;
// out o[k] = {a.byte k, b.byte k, c.byte k, d.byte k} (pixel 0 in the low byte).
__device__ __forceinline__ void transpose4x4(uint32_t a, uint32_t b, uint32_t c,
                                             uint32_t d, uint32_t &o0, uint32_t &o1,
                                             uint32_t &o2, uint32_t &o3)
{
    uint32_t t0 = perm(b, a, 0x05010400u);  // a0 b0 a1 b1
    uint32_t t1 = perm(b, a, 0x07030602u);  // a2 b2 a3 b3
    uint32_t t2 = perm(d, c, 0x05010400u);  // c0 d0 c1 d1
    uint32_t t3 = perm(d, c, 0x07030602u);  // c2 d2 c3 d3
    o0 = perm(t2, t0, 0x05040100u);         // a0 b0 c0 d0
    o1 = perm(t2, t0, 0x07060302u);         // a1 b1 c1 d1
    o2 = perm(t3, t1, 0x05040100u);         // a2 b2 c2 d2
    o3 = perm(t3, t1, 0x07060302u);         // a3 b3 c3 d3
}

__device__ __forceinline__ uint32_t gather_byte0(uint32_t a, uint32_t b, uint32_t c,
                                                 uint32_t d)
{
    uint32_t t0 = perm(b, a, 0x0c0c0400u);  // a0 b0 0 0
    uint32_t t1 = perm(d, c, 0x04000c0cu);  // 0 0 c0 d0
    return t0 | t1;
}

__device__ __forceinline__ u32x4 load16_aligned_nt(const uint8_t *p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
}

__device__ __forceinline__ u32x4 load16_aligned(const uint8_t *p)
{
    return *reinterpret_cast<const u32x4 *>(p);
}

__device__ __forceinline__ u32x4 load16_any(const uint8_t *p)
{
    // gfx950 runs with unaligned global access enabled; tell the compiler the
    // pointer is only byte aligned and let it pick the widest legal load.
    typedef u32x4 u32x4_u __attribute__((aligned(1)));
    return *reinterpret_cast<const u32x4_u *>(p);
}

template <bool NT = true>
__device__ __forceinline__ void store16(uint8_t *p, u32x4 v)
{
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else
        *reinterpret_cast<u32x4 *>(p) = v;
}

// Coarse soil row of strip row y.  cj comes from the caller (host-built, already
// clamped as src/cn.c:229 does); the extra min() only keeps a bad map from
// reading outside the workspace.
__device__ __forceinline__ uint32_t soil_row(const StripParams &p, uint32_t y)
{
    const uint32_t r = (uint32_t)p.cj[y];
    return r < p.hx_rows ? r : p.hx_rows - 1u;
}

__device__ __forceinline__ uint32_t clamp_row(const StripParams &p, int32_t r)
{
    const uint32_t u = (uint32_t)r;
    return u < p.hx_rows ? u : p.hx_rows - 1u;
}

// cj[i] for a wave-uniform i through the scalar data cache (s_load_dword):
// constant-address-space loads with a uniform address select SMEM, which has
// its own counter (lgkmcnt) and does not queue behind vector memory traffic.
// cj is written by a copy that completed before this kernel started.
__device__ __forceinline__ int32_t scalar_load_i32(const int32_t *base, uint32_t i)
{
    typedef const int32_t __attribute__((address_space(4))) *const_ptr_t;
    const_ptr_t cp = (const_ptr_t)(uintptr_t)base;
    return cp[__builtin_amdgcn_readfirstlane(i)];
}

// Workgroup -> chunk mapping.  Workgroups are dealt round-robin over the 8
// XCDs (b and b+8 share one L2), so give each XCD a contiguous slab of the
// strip: the x-expanded soil rows a slab re-reads then stay in that XCD's L2.
__device__ __forceinline__ uint32_t first_chunk(uint32_t nchunks, uint32_t &step,
                                                uint32_t &end, bool xcd_slabs)
{
    uint32_t nb = gridDim.x;
    uint32_t b = blockIdx.x;
    if (xcd_slabs && nb % 8u == 0u && nchunks >= nb) {
        uint32_t xcd = b & 7u;
        uint32_t per = (nchunks + 7u) / 8u;
        uint32_t lo = xcd * per;
        uint32_t hi = lo + per < nchunks ? lo + per : nchunks;
        step = nb / 8u;
        end = hi;
        return lo + (b >> 3);
    }
    step = nb;
    end = nchunks;
    return b;
}

// ------------------------------------------------------------------------
// cn_strip: the fused block kernel.
//
// KIND = kLut16 (all-tables): one 16-byte LDS row per (soil plane, class) holds
//   the CN of all nine tables, so a pixel costs one ds_read_b128 per drainage
//   condition whatever the number of rasters written; 4x4 byte transposes
//   (v_perm_b32) turn "9 values of one pixel" into "4 pixels of one raster".
// KIND = kLut1 (single table, BASELINE config 2): 2 B/px of HBM traffic, so the
//   pixel rate is ~8x higher and the LDS work per pixel is one byte read.
// ILP  = 1024-px wave chunks a wave handles per loop trip; all their loads are
//   issued before any is consumed.
// NT   = nontemporal landcover loads and raster stores.
// PF   = software pipeline: the loads of trip t+1 are issued before trip t's
//   gathers and stores, on two register sets used alternately (vmcnt counts
//   loads and stores in one in-order queue on gfx950, so a wave that loads,
//   stores, loads ... has nothing in flight while it gathers; with the next
//   trip's loads issued first, its stores and those loads overlap).
//
// The loop body has no divergent region: a raster is processed as a flat byte
// array in 16-byte lane groups, every lane of every wave issues the same loads,
// and the three irregular cases are folded in without a branch that holds
// memory operations --
//   * lanes past the last full 16-byte group load from offset 0 and skip the
//     store (exec mask on the store only); the npix % 16 tail pixels are done
//     byte-wise after the loop;
//   * lanes behind a row end inside the wave's 1024-px span use the next
//     raster row's soil row (both coarse rows come through the scalar cache);
//   * only when W % 16 != 0 can a lane's 16 pixels straddle a row end: the
//     wave that holds such a lane (a scalar test, one wave in ~35) issues a
//     second soil load for it that starts n bytes before the next soil row, and
//     merges the two vectors with byte masks.
// Requires W >= kMinVectorW so that a wave's span wraps at most once.
// ------------------------------------------------------------------------
enum { kLut16 = 0, kLut1 = 1 };
constexpr uint32_t kWavePx = 64u * kPxPerLane;      // 1024 px per wave chunk
constexpr uint32_t kMinVectorW = kWavePx + 16u;

// What a lane holds of one trip between issuing its loads and using them.
template <int ILP>
struct Trip {
    u32x4 e16[ILP], c16[ILP];
    uint32_t i0[ILP];
};

template <int ILP, bool NT, bool HX4>
__device__ __forceinline__ void issue_trip(const StripParams &p, uint32_t trip, uint32_t lane_off,
                                           uint32_t wave_off, Trip<ILP> &tr)
{
    // ---- wave-uniform part: row and column of the wave's first pixel, the coarse rows of that
    // raster row and of the next one through the scalar cache (s_load, own counter) ----
    uint32_t xb[ILP], r0[ILP], r1[ILP], wb[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        wb[u] = __builtin_amdgcn_readfirstlane((trip * ILP + u) * (uint32_t)kChunk + wave_off);
        const uint32_t wbc = wb[u] < p.npix ? wb[u] : 0u;   // a wave past the end only needs valid addresses
        const uint32_t y = wbc / p.W;
        xb[u] = wbc - y * p.W;
        r0[u] = (uint32_t)scalar_load_i32(p.cj, y);
        r1[u] = (uint32_t)scalar_load_i32(p.cj, y + 1u < p.rows ? y + 1u : y);
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        tr.i0[u] = wb[u] + lane_off;
        const uint8_t *pe = p.esa + (tr.i0[u] < p.nvec16 ? tr.i0[u] : 0u);
        tr.e16[u] = NT ? load16_aligned_nt(pe) : load16_aligned(pe);
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        uint32_t xl = xb[u] + lane_off;
        const bool wrap = xl >= p.W;
        xl = wrap ? xl - p.W : xl;
        const uint32_t row = clamp_row(p, (int32_t)(wrap ? r1[u] : r0[u]));
#if defined(GCN10_DIAG) && (GCN10_DIAG == 2 || GCN10_DIAG == 3)
        tr.c16[u] = u32x4{ row, xl, 0u, 0u } & 0x11111111u;    // timing-only build: no soil load
#else
        if (HX4) {
            // W % 16 == 0: the lane's 16 pixels are one column group; its soil codes are one dword
            tr.c16[u] = u32x4{ p.hx4[(size_t)row * p.hx4_stride + (xl >> 4)], 0u, 0u, 0u };
            continue;
        }
        const uint8_t *pa = p.hx + (size_t)row * p.hx_stride + xl;
        u32x4 a = load16_any(pa);
        if (p.W & 15u) {
            const uint32_t n = p.W - xl;            // pixels of this lane that are still in its row
            const bool strad = !wrap && n < (uint32_t)kPxPerLane;
            if (__builtin_amdgcn_ballot_w64(strad) != 0ull) {
                // the other lanes of this wave re-read their own vector (an L1 hit)
                const uint8_t *pb = strad ? p.hx + (size_t)clamp_row(p, (int32_t)r1[u]) * p.hx_stride - n : pa;
                const u32x4 b = load16_any(pb);
                const uint32_t nn = strad ? n : (uint32_t)kPxPerLane;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int32_t m = (int32_t)nn - 4 * j;      // bytes of dword j that come from a
                    const uint32_t mask = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : (1u << (8 * m)) - 1u);
                    a[j] = (a[j] & mask) | (b[j] & ~mask);
                }
            }
        }
        tr.c16[u] = a;
#endif
    }
}

// Table gathers and stores of one trip.
// Soil code bytes of a lane's 16 pixels from the compact word {code a, code b, split, -}: pixels [0, split)
// have code a, the rest code b.
__device__ __forceinline__ u32x4 expand_soil_word(uint32_t w)
{
    const uint32_t a = (w & 0xffu) * 0x01010101u, b = ((w >> 8) & 0xffu) * 0x01010101u;
    const int32_t split = (int32_t)((w >> 16) & 0xffu);
    u32x4 cd;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int32_t m = split - 4 * j;        // bytes of dword j that hold code a
        const uint32_t mask = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : (1u << (8 * m)) - 1u);
        cd[j] = (a & mask) | (b & ~mask);
    }
    return cd;
}

template <int KIND, int COND_MASK, int ILP, bool NT, bool HX4>
__device__ __forceinline__ void finish_trip(const StripParams &p, const uint8_t *lut, uint32_t tmask,
                                            const Trip<ILP> &tr)
{
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        const bool live = tr.i0[u] < p.nvec16;
        const u32x4 c16 = HX4 ? expand_soil_word(tr.c16[u][0]) : tr.c16[u];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (!(COND_MASK & (1 << c)))
                continue;
            if (KIND == kLut16) {
                uint32_t acc[9][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t e = tr.e16[u][j];
                    const uint32_t cd = c16[j];
                    u32x4 r4[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lc16 = q == 0 ? (e << 4) & 0xff0u : (e >> (8 * q - 4)) & 0xff0u;
                        const uint32_t s = (cd >> (8 * q + 4 * c)) & 0xfu;
                        const uint32_t addr = s * (uint32_t)kPlane16 + lc16;
                        r4[q] = *reinterpret_cast<const u32x4 *>(lut + addr);
                    }
                    transpose4x4(r4[0][0], r4[1][0], r4[2][0], r4[3][0], acc[0][j], acc[1][j],
                                 acc[2][j], acc[3][j]);
                    transpose4x4(r4[0][1], r4[1][1], r4[2][1], r4[3][1], acc[4][j], acc[5][j],
                                 acc[6][j], acc[7][j]);
                    acc[8][j] = gather_byte0(r4[0][2], r4[1][2], r4[2][2], r4[3][2]);
                }
                if (live) {
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        if (tmask & (1u << k)) {
                            u32x4 v = {acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
                            store16<NT>(p.out[c * 9 + k] + tr.i0[u], v);
                        }
                    }
                }
            }
            else {
                u32x4 v;
#if defined(GCN10_DIAG) && (GCN10_DIAG == 1 || GCN10_DIAG == 3)
                v = tr.e16[u] ^ c16;          // timing-only build: no table lookup
#else
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t e = tr.e16[u][j];
                    const uint32_t cd = c16[j];
                    uint32_t w = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lc = (e >> (8 * q)) & 0xffu;
                        const uint32_t s = (cd >> (8 * q + 4 * c)) & 0xfu;
                        const uint32_t b = lut[s * (uint32_t)kPlane1 + lc];
                        w |= b << (8 * q);
                    }
                    v[j] = w;
                }
#endif
                if (live)
                    store16<NT>(p.out[c * 9 + p.single_k] + tr.i0[u], v);
            }
        }
    }
}

// The trip loop of cn_strip_kernel (HX4: soil codes from the compact words).
template <int KIND, int COND_MASK, int ILP, bool NT, bool PF, bool HX4>
__device__ __forceinline__ void strip_loop(const StripParams &p, const uint8_t *lut, uint32_t tmask,
                                           uint32_t lane_off, uint32_t wave_off)
{
    // p.nchunks counts trips (groups of ILP chunks of 4096 px) here
    uint32_t step, end;
    uint32_t trip = first_chunk(p.nchunks, step, end, p.xcd_slabs != 0);
    if (!PF) {
        for (; trip < end; trip += step) {
            Trip<ILP> tr;
            issue_trip<ILP, NT, HX4>(p, trip, lane_off, wave_off, tr);
            finish_trip<KIND, COND_MASK, ILP, NT, HX4>(p, lut, tmask, tr);
        }
    }
    else if (trip < end) {
        Trip<ILP> ta, tb;       // used alternately: a copy would wait for the loads it copies
        issue_trip<ILP, NT, HX4>(p, trip, lane_off, wave_off, ta);
        for (;;) {
            trip += step;
            if (trip >= end) {
                finish_trip<KIND, COND_MASK, ILP, NT, HX4>(p, lut, tmask, ta);
                break;
            }
            issue_trip<ILP, NT, HX4>(p, trip, lane_off, wave_off, tb);
            finish_trip<KIND, COND_MASK, ILP, NT, HX4>(p, lut, tmask, ta);
            trip += step;
            if (trip >= end) {
                finish_trip<KIND, COND_MASK, ILP, NT, HX4>(p, lut, tmask, tb);
                break;
            }
            issue_trip<ILP, NT, HX4>(p, trip, lane_off, wave_off, ta);
            finish_trip<KIND, COND_MASK, ILP, NT, HX4>(p, lut, tmask, tb);
        }
    }
}

template <int KIND, int COND_MASK, bool ALL_TABLES, int ILP, bool NT, bool PF>
__global__ __launch_bounds__(kThreads) void cn_strip_kernel(const StripParams p)
{
    constexpr int kLutBytes = KIND == kLut16 ? kLut16Bytes : kLut1Bytes;
    __shared__ __attribute__((aligned(16))) uint8_t lut[kLutBytes];

    if (KIND == kLut16) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.lut);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lut);
        for (int i = threadIdx.x; i < kLutBytes / 16; i += kThreads)
            dst[i] = src[i];
    }
    else {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.lut);
        uint32_t *dst = reinterpret_cast<uint32_t *>(lut);
        for (int i = threadIdx.x; i < kLutBytes / 4; i += kThreads)
            dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t lane_off = (threadIdx.x & 63u) * kPxPerLane;
    const uint32_t wave_off = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kWavePx;
    const uint32_t tmask = ALL_TABLES ? 0x1ffu : p.table_mask;

    // compact soil words when the host offers them and no column group of this tile is complex
    // (wave-uniform: the flag comes through the scalar cache)
    if (p.hx4 && (uint32_t)scalar_load_i32(reinterpret_cast<const int32_t *>(p.hx4_complex), 0u) != p.hx4_gen)
        strip_loop<KIND, COND_MASK, ILP, NT, PF, true>(p, lut, tmask, lane_off, wave_off);
    else
        strip_loop<KIND, COND_MASK, ILP, NT, PF, false>(p, lut, tmask, lane_off, wave_off);

    // the last npix % 16 pixels, one per thread
    if (blockIdx.x == 0 && threadIdx.x < p.npix - p.nvec16) {
        const uint32_t i = p.nvec16 + threadIdx.x;
        const uint32_t y = i / p.W;
        const uint32_t x = i - y * p.W;
        const uint32_t lc = p.esa[i];
        const uint32_t cd = p.hx[(size_t)soil_row(p, y) * p.hx_stride + x];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (!(COND_MASK & (1 << c)))
                continue;
            const uint32_t s = (cd >> (4 * c)) & 0xfu;
            if (KIND == kLut16) {
                const uint8_t *row = lut + s * (uint32_t)kPlane16 + lc * 16u;
                for (int k = 0; k < 9; k++)
                    if (tmask & (1u << k))
                        p.out[c * 9 + k][i] = row[k];
            }
            else {
                p.out[c * 9 + p.single_k][i] = lut[s * (uint32_t)kPlane1 + lc];
            }
        }
    }
}

// ------------------------------------------------------------------------
// plain 1R:1W copy with the strip kernel's launch shape (XCD slabs, 16 B per
// lane, two chunks per trip, nontemporal): what this device streams when a
// kernel does nothing but move the bytes.  bench.py times it in the same run
// as the strip kernel (gcn10_gpu_stream_copy).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void stream_copy_kernel(const u32x4 *in, u32x4 *out, uint32_t nvec,
                                                               uint32_t ntrips)
{
    uint32_t step, end;
    for (uint32_t trip = first_chunk(ntrips, step, end, true); trip < end; trip += step) {
        u32x4 v[2];
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (trip * 2u + u) * (uint32_t)kThreads + threadIdx.x;
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0u));
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
    }
}

// ------------------------------------------------------------------------
// byte-wise strip kernel: any alignment, any shape.  Used when a raster
// pointer is not 16-byte aligned (never the case for the host pipeline).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void cn_strip_bytes(const StripParams p,
                                                           uint32_t cond_mask)
{
    __shared__ __attribute__((aligned(16))) uint8_t lut[kLut16Bytes];
    for (int i = threadIdx.x; i < kLut16Bytes; i += kThreads)
        lut[i] = p.lut[i];
    __syncthreads();

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.npix; i += stride) {
        const uint32_t y = i / p.W;
        const uint32_t x = i - y * p.W;
        const uint32_t lc = p.esa[i];
        const uint32_t cd = p.hx[(size_t)soil_row(p, y) * p.hx_stride + x];
        for (int c = 0; c < 2; c++) {
            if (!(cond_mask & (1u << c)))
                continue;
            const uint32_t s = (cd >> (4 * c)) & 0xfu;
            const uint8_t *row = lut + s * (uint32_t)kPlane16 + lc * 16u;
            for (int k = 0; k < 9; k++)
                if (p.table_mask & (1u << k))
                    p.out[c * 9 + k][i] = row[k];
        }
    }
}

// ------------------------------------------------------------------------
// x-expansion of the coarse soil window (the x half of src/cn.c:218-232):
// hx[r][x] = soil_code(coarse[r][ci[x]]) for every coarse row r.
//
// Next to the bytes, one COMPACT WORD per 16-px column group and coarse row:
// hx4[r][g] = code a | code b << 8 | split << 16, meaning pixels [0, split)
// of the group have code a and the rest code b.  At the usual ratio (25 fine
// columns per coarse cell) every group has that form, and the strip kernels
// then load one dword per lane instead of 16 bytes (the soil stream costs the
// single-raster kernel 5 % with bytes, 1-2 % with words: DESIGN.md section 5).
// A group that has no such form (three cells under 16 columns, a map that is
// not monotone) gets split = 0xff and stores this tile's generation number in
// *complex (no reset between tiles needed); the strip kernels compare that
// word with the generation they were launched for and use the bytes for the
// whole tile when it matches.
// ------------------------------------------------------------------------
#ifndef GCN10_EXPAND_ROWS
#define GCN10_EXPAND_ROWS 4     // 2: 20.0 us, 3: 17.3, 4: 16.5, 6: 17.6 for a 36000 x 1440 window (profiles/r02/prepare_tile_rows_per_thread.txt)
#endif
constexpr uint32_t kExpandRows = GCN10_EXPAND_ROWS;     // coarse rows per thread of expand_x_codes

template <bool VEC>
__global__ __launch_bounds__(kThreads) void expand_x_codes(const uint8_t *coarse,
                                                           uint32_t hsx, uint32_t hsy,
                                                           const int32_t *ci, uint32_t W,
                                                           uint8_t *hx, uint32_t hx_stride,
                                                           uint32_t *hx4, uint32_t *complex, uint32_t gen)
{
    // one thread = 16 consecutive fine columns of kExpandRows coarse rows: the 16 column indices are
    // loaded once (dwordx4 when ci is 16-byte aligned) and clamped once, then per row 16 byte gathers
    // that hit L1/L2 (a coarse row is ~1.4 KB) and one 16-byte store
    const uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    const uint32_t r0 = blockIdx.y * kExpandRows;
    if (x >= hx_stride || r0 >= hsy)
        return;
    const uint8_t pad = (uint8_t)(kInvalidPlane | (kInvalidPlane << 4));
    uint32_t cx[16];
    if (VEC && x + 16u <= W) {
        const u32x4 *civ = reinterpret_cast<const u32x4 *>(ci + x);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32x4 c4 = civ[j];
#pragma unroll
            for (int q = 0; q < 4; q++)
                cx[4 * j + q] = c4[q] < hsx ? c4[q] : hsx - 1u;
        }
    }
    else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            // columns past W are padding: marked here, given the "invalid" code below
            const uint32_t xx = x + q;
            // (an index of the raster is clamped exactly as in the vector path, whatever its value;
            // only columns past W get the padding mark)
            const uint32_t c = xx < W ? (uint32_t)ci[xx] : 0u;
            cx[q] = xx < W ? (c < hsx ? c : hsx - 1u) : 0xffffffffu;
        }
    }
    // at the usual ratio (25 fine columns per coarse cell) 16 consecutive columns see at most two coarse
    // cells, the first column's and the last one's: two byte loads and 16 selects instead of 16 byte
    // gathers (the kernel is bound by the number of load instructions, not by bytes)
    const uint32_t c_lo = cx[0], c_hi = cx[15];
    const uint32_t r1 = r0 + kExpandRows < hsy ? r0 + kExpandRows : hsy;
    // compact form: a run of c_lo followed by a run of c_hi (also: padding only, right of the raster)
    uint32_t split = 0;
    bool compact = true, in_set = true;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const bool lo = cx[q] == c_lo;
        in_set = in_set && (lo || cx[q] == c_hi);
        compact = compact && (lo ? split == (uint32_t)q : cx[q] == c_hi);
        split += lo && split == (uint32_t)q ? 1u : 0u;
    }
    const bool two = c_lo != 0xffffffffu && c_hi != 0xffffffffu && in_set;
    if (hx4 && x < W && !compact && *reinterpret_cast<volatile uint32_t *>(complex) != gen)
        *reinterpret_cast<volatile uint32_t *>(complex) = gen;      // every writer stores the same value
    for (uint32_t r = r0; r < r1; r++) {
        const uint8_t *row = coarse + (size_t)r * hsx;
        u32x4 o;
        if (two) {
            const uint32_t code_lo = soil_code(row[c_lo]), code_hi = soil_code(row[c_hi]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    w |= (cx[4 * j + q] == c_lo ? code_lo : code_hi) << (8 * q);
                o[j] = w;
            }
        }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t c = cx[4 * j + q];
                    const uint8_t code = c == 0xffffffffu ? pad : soil_code(row[c]);
                    w |= (uint32_t)code << (8 * q);
                }
                o[j] = w;
            }
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(hx + (size_t)r * hx_stride + x));
        if (hx4)
            __builtin_nontemporal_store(compact ? (o[0] & 0xffu) | ((o[3] >> 24) << 8) | (split << 16) : 0x00ff0000u,
                                        hx4 + (size_t)r * (hx_stride / 16u) + x / 16u);
    }
}

__global__ __launch_bounds__(kThreads) void resample_rows(const uint8_t *coarse,
                                                          uint32_t hsx,
                                                          const int32_t *ci,
                                                          const int32_t *cj, uint32_t W,
                                                          uint32_t rows, uint8_t *out)
{
    const size_t npix = (size_t)W * rows;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const uint32_t y = (uint32_t)(i / W);
        const uint32_t x = (uint32_t)(i - (size_t)y * W);
        out[i] = coarse[(size_t)(uint32_t)cj[y] * hsx + (uint32_t)ci[x]];   // src/cn.c:230
    }
}

// ------------------------------------------------------------------------
// modify_hysogs_data (src/cn.c:88-111), in place
// ------------------------------------------------------------------------
__device__ __forceinline__ uint8_t remap_soil(uint8_t h, bool drained)
{
    if (h >= 11 && h <= 14)
        return drained ? (uint8_t)4 : (uint8_t)(h - 10);
    return h;
}

__global__ __launch_bounds__(kThreads) void modify_hysogs_kernel(uint8_t *h, size_t npix,
                                                                 int drained,
                                                                 size_t head, size_t nvec)
{
    // [0, head) bytes, then nvec aligned 16-byte groups, then the tail bytes
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 *body = reinterpret_cast<u32x4 *>(h + head);
    for (size_t v = tid; v < nvec; v += stride) {
        u32x4 in = body[v];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t w = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                w |= (uint32_t)remap_soil((uint8_t)(in[j] >> (8 * q)), drained != 0) << (8 * q);
            o[j] = w;
        }
        body[v] = o;
    }
    const size_t tail0 = head + nvec * 16;
    const size_t nscalar = head + (npix - tail0);
    for (size_t s = tid; s < nscalar; s += stride) {
        const size_t i = s < head ? s : tail0 + (s - head);
        h[i] = remap_soil(h[i], drained != 0);
    }
}

// ------------------------------------------------------------------------
// calculate_cn (src/cn.c:114-131) on a full-resolution soil raster
// ------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void calculate_cn_kernel(const uint8_t *esa,
                                                                const uint8_t *hsg,
                                                                size_t npix,
                                                                const uint8_t *lut_img,
                                                                uint8_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lut[kLut1Bytes];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lut_img);
        uint32_t *dst = reinterpret_cast<uint32_t *>(lut);
        for (int i = threadIdx.x; i < kLut1Bytes / 4; i += kThreads)
            dst[i] = src[i];
    }
    __syncthreads();

    const size_t nvec = VEC ? npix / 16 : 0;
    if (VEC) {
        // chunks of 2 x 256 vectors; every XCD streams a contiguous eighth (first_chunk)
        constexpr uint32_t kPer = 2;
        const uint32_t nchunks = (uint32_t)((nvec + kPer * kThreads - 1) / (kPer * kThreads));
        uint32_t step, end;
        for (uint32_t chunk = first_chunk(nchunks, step, end, true); chunk < end; chunk += step) {
            size_t v[kPer];
            u32x4 e16[kPer], h16[kPer];
            bool ok[kPer];
#pragma unroll
            for (uint32_t u = 0; u < kPer; u++) {
                v[u] = ((size_t)chunk * kPer + u) * kThreads + threadIdx.x;
                ok[u] = v[u] < nvec;
                const size_t at = ok[u] ? v[u] : 0;
                e16[u] = load16_aligned_nt(esa + at * 16);
                h16[u] = load16_aligned_nt(hsg + at * 16);
            }
#pragma unroll
            for (uint32_t u = 0; u < kPer; u++) {
                if (!ok[u])
                    continue;
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t w = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lc = (e16[u][j] >> (8 * q)) & 0xffu;
                        uint32_t s = (h16[u][j] >> (8 * q)) & 0xffu;
                        s = s < 5u ? s : (uint32_t)kInvalidPlane;   // src/cn.c:123-124
                        w |= (uint32_t)lut[s * (uint32_t)kPlane1 + lc] << (8 * q);
                    }
                    o[j] = w;
                }
                store16(out + v[u] * 16, o);
            }
        }
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = nvec * 16 + tid; i < npix; i += stride) {
        const uint32_t lc = esa[i];
        uint32_t s = hsg[i];
        s = s < 5u ? s : (uint32_t)kInvalidPlane;
        out[i] = lut[s * (uint32_t)kPlane1 + lc];
    }
}

inline bool aligned16(const void *p)
{
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

inline int popcount(unsigned v)
{
    return __builtin_popcount(v);
}

typedef void (*strip_kernel_t)(const StripParams);

template <int KIND, int ILP, bool NT, bool PF>
strip_kernel_t pick_by_mask2(unsigned cond_mask, bool all)
{
    // the single-table kernel ignores ALL_TABLES: instantiate it once
    constexpr bool kSingle = KIND == kLut1;
    switch (cond_mask) {
    case 1:
        return (all || kSingle) ? cn_strip_kernel<KIND, 1, true, ILP, NT, PF>
                                : cn_strip_kernel<KIND, 1, kSingle, ILP, NT, PF>;
    case 2:
        return (all || kSingle) ? cn_strip_kernel<KIND, 2, true, ILP, NT, PF>
                                : cn_strip_kernel<KIND, 2, kSingle, ILP, NT, PF>;
    default:
        return (all || kSingle) ? cn_strip_kernel<KIND, 3, true, ILP, NT, PF>
                                : cn_strip_kernel<KIND, 3, kSingle, ILP, NT, PF>;
    }
}

// (the pipeline choice `pf` is an argument all the way down: two worker threads pick kernels for their
// own contexts at the same time, a process-wide flag would be a data race)
template <int KIND, int ILP, bool NT>
strip_kernel_t pick_by_mask(unsigned cond_mask, bool all, bool pf)
{
    // pipelined variants exist for ILP 1 and 2 (two register sets of ILP 4 cost occupancy)
    if (ILP <= 2 && pf)
        return pick_by_mask2<KIND, (ILP <= 2 ? ILP : 1), NT, true>(cond_mask, all);
    return pick_by_mask2<KIND, ILP, NT, false>(cond_mask, all);
}

template <int KIND>
strip_kernel_t pick_by_ilp(unsigned cond_mask, bool all, int ilp, bool nt, bool pf)
{
    switch (ilp) {
    case 1: return nt ? pick_by_mask<KIND, 1, true>(cond_mask, all, pf) : pick_by_mask<KIND, 1, false>(cond_mask, all, pf);
    case 2: return nt ? pick_by_mask<KIND, 2, true>(cond_mask, all, pf) : pick_by_mask<KIND, 2, false>(cond_mask, all, pf);
    case 4:
        if (KIND == kLut1)
            return nt ? pick_by_mask<kLut1, 4, true>(cond_mask, all, pf) : pick_by_mask<kLut1, 4, false>(cond_mask, all, pf);
        return nullptr;
    default: return nullptr;
    }
}

strip_kernel_t pick_strip_kernel(bool single, unsigned cond_mask, bool all, int ilp, bool nt, bool pf)
{
    return single ? pick_by_ilp<kLut1>(cond_mask, all, ilp, nt, pf)
                  : pick_by_ilp<kLut16>(cond_mask, all, ilp, nt, pf);
}

}  // namespace

namespace gcn10 {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int use_device(gcn10_gpu_ctx *ctx)
{
    if (!ctx)
        return fail(GCN10_E_INVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return GCN10_OK;
}

hipStream_t as_stream(gcn10_gpu_ctx *ctx, gcn10_stream_t s)
{
    return s ? reinterpret_cast<hipStream_t>(s) : ctx->main_stream;
}

}  // namespace gcn10

namespace {

// Memory-bound streaming grid: a few workgroups per CU, each looping over
// chunks; a multiple of 8 so every XCD gets the same number.
uint32_t stream_grid(const gcn10_gpu_ctx *ctx, uint64_t nchunks, int blocks_per_cu = 0)
{
    uint64_t cap = (uint64_t)ctx->n_cus * (blocks_per_cu > 0 ? blocks_per_cu : ctx->grid_blocks_per_cu);
    cap -= cap % 8u;
    if (cap < 8)
        cap = 8;
    uint64_t g = nchunks < cap ? nchunks : cap;
    return (uint32_t)(g ? g : 1);
}

}  // namespace

// ------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------
extern "C" {

int gcn10_gpu_abi_version(void)
{
    return GCN10_GPU_ABI_VERSION;
}

const char *gcn10_gpu_last_error(void)
{
    return gcn10::g_err;
}

int gcn10_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int gcn10_gpu_init(int device, gcn10_gpu_ctx **out)
{
    if (!out)
        return fail(GCN10_E_INVAL, "gcn10_gpu_init: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GCN10_E_NODEVICE, "no HIP device visible (%s); this engine has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n)
        return fail(GCN10_E_INVAL, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GCN10_E_NODEVICE, "device %d is %s; this library carries gfx950 code only",
                    device, prop.gcnArchName);

    gcn10_gpu_ctx *ctx = new (std::nothrow) gcn10_gpu_ctx();
    if (!ctx)
        return fail(GCN10_E_NOMEM, "context allocation failed");
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount;
    hipError_t e2 = hipStreamCreateWithFlags(&ctx->main_stream, hipStreamNonBlocking);
    if (e2 == hipSuccess)
        e2 = hipMalloc(reinterpret_cast<void **>(&ctx->d_lut16), kLut16Bytes);
    if (e2 == hipSuccess)
        e2 = hipMalloc(reinterpret_cast<void **>(&ctx->d_lut1), GCN10_N_TABLES * kLut1Bytes);
    if (e2 != hipSuccess) {
        int rc = fail(GCN10_E_HIP, "context setup: %s", hipGetErrorString(e2));
        gcn10_gpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return GCN10_OK;
}

void gcn10_gpu_destroy(gcn10_gpu_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->main_stream) {
        (void)hipStreamSynchronize(ctx->main_stream);
        (void)hipStreamDestroy(ctx->main_stream);
    }
    if (ctx->d_lut16)
        (void)hipFree(ctx->d_lut16);
    if (ctx->d_lut1)
        (void)hipFree(ctx->d_lut1);
    if (ctx->d_hx_alloc)
        (void)hipFree(ctx->d_hx_alloc);
    if (ctx->deflate_ws)
        (void)hipFree(ctx->deflate_ws);
    if (ctx->inflate_ws)
        (void)hipFree(ctx->inflate_ws);
    if (ctx->d_class_of)
        (void)hipFree(ctx->d_class_of);
    delete ctx;
}

int gcn10_gpu_device_info(gcn10_gpu_ctx *ctx, char *name, size_t cap, size_t *hbm_bytes)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name && cap)
        snprintf(name, cap, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName);
    if (hbm_bytes)
        *hbm_bytes = prop.totalGlobalMem;
    return prop.multiProcessorCount;
}

int gcn10_gpu_pci_bus_id(int device, char *buf, size_t cap)
{
    if (!buf || cap < 13)
        return fail(GCN10_E_INVAL, "gcn10_gpu_pci_bus_id: buffer too small");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)cap, device));
    return GCN10_OK;
}

// ---- memory / streams / events ------------------------------------------

int gcn10_gpu_malloc(gcn10_gpu_ctx *ctx, size_t bytes, void **dptr)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!dptr)
        return fail(GCN10_E_INVAL, "gcn10_gpu_malloc: null out pointer");
    *dptr = nullptr;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return GCN10_OK;
}

int gcn10_gpu_free(gcn10_gpu_ctx *ctx, void *dptr)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (dptr)
        HIP_TRY(hipFree(dptr));
    return GCN10_OK;
}

int gcn10_gpu_host_alloc(gcn10_gpu_ctx *ctx, size_t bytes, void **hptr)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!hptr)
        return fail(GCN10_E_INVAL, "gcn10_gpu_host_alloc: null out pointer");
    *hptr = nullptr;
    HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return GCN10_OK;
}

int gcn10_gpu_host_free(gcn10_gpu_ctx *ctx, void *hptr)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (hptr)
        HIP_TRY(hipHostFree(hptr));
    return GCN10_OK;
}

int gcn10_gpu_memcpy_h2d(gcn10_gpu_ctx *ctx, void *dst, const void *src, size_t bytes,
                         gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (bytes && (!dst || !src))
        return fail(GCN10_E_INVAL, "gcn10_gpu_memcpy_h2d: null pointer");
    if (bytes)
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(ctx, stream)));
    return GCN10_OK;
}

int gcn10_gpu_memcpy_d2h(gcn10_gpu_ctx *ctx, void *dst, const void *src, size_t bytes,
                         gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (bytes && (!dst || !src))
        return fail(GCN10_E_INVAL, "gcn10_gpu_memcpy_d2h: null pointer");
    if (bytes)
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(ctx, stream)));
    return GCN10_OK;
}

int gcn10_gpu_memset(gcn10_gpu_ctx *ctx, void *dptr, int value, size_t bytes,
                     gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (bytes && !dptr)
        return fail(GCN10_E_INVAL, "gcn10_gpu_memset: null pointer");
    if (bytes)
        HIP_TRY(hipMemsetAsync(dptr, value, bytes, as_stream(ctx, stream)));
    return GCN10_OK;
}

int gcn10_gpu_stream_create(gcn10_gpu_ctx *ctx, gcn10_stream_t *stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!stream)
        return fail(GCN10_E_INVAL, "gcn10_gpu_stream_create: null out pointer");
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return GCN10_OK;
}

int gcn10_gpu_stream_destroy(gcn10_gpu_ctx *ctx, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (stream)
        HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return GCN10_OK;
}

int gcn10_gpu_stream_sync(gcn10_gpu_ctx *ctx, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(as_stream(ctx, stream)));
    return GCN10_OK;
}

int gcn10_gpu_device_sync(gcn10_gpu_ctx *ctx)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    return GCN10_OK;
}

int gcn10_gpu_event_create(gcn10_gpu_ctx *ctx, gcn10_event_t *ev)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_event_create: null out pointer");
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return GCN10_OK;
}

int gcn10_gpu_event_destroy(gcn10_gpu_ctx *ctx, gcn10_event_t ev)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (ev)
        HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return GCN10_OK;
}

int gcn10_gpu_event_record(gcn10_gpu_ctx *ctx, gcn10_event_t ev, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_event_record: null event");
    HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), as_stream(ctx, stream)));
    return GCN10_OK;
}

int gcn10_gpu_event_sync(gcn10_gpu_ctx *ctx, gcn10_event_t ev)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_event_sync: null event");
    if (ctx->event_sync_sleep_us > 0) {
        // a waiter that sleeps between queries: hipEventSynchronize spins in user mode for the whole wait, which is
        // what a timing loop wants and what a thread of a CPU-quota-bound host pipeline does not (the threads that
        // wait for a strip's bytes burnt 18 % of the program's CPU time: profiles/r03/host_cpu_by_thread_and_job.txt)
        for (;;) {
            const hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(ev));
            if (e == hipSuccess)
                break;
            if (e != hipErrorNotReady)
                HIP_TRY(e);
            usleep((useconds_t)ctx->event_sync_sleep_us);
        }
        return GCN10_OK;
    }
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev)));
    return GCN10_OK;
}

int gcn10_gpu_stream_wait_event(gcn10_gpu_ctx *ctx, gcn10_stream_t stream, gcn10_event_t ev)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_stream_wait_event: null event");
    HIP_TRY(hipStreamWaitEvent(as_stream(ctx, stream), reinterpret_cast<hipEvent_t>(ev), 0));
    return GCN10_OK;
}

int gcn10_gpu_event_elapsed_ms(gcn10_gpu_ctx *ctx, gcn10_event_t start, gcn10_event_t stop,
                               float *ms)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!start || !stop || !ms)
        return fail(GCN10_E_INVAL, "gcn10_gpu_event_elapsed_ms: null argument");
    HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start),
                                reinterpret_cast<hipEvent_t>(stop)));
    return GCN10_OK;
}

// ---- tables ---------------------------------------------------------------

int gcn10_gpu_set_tables(gcn10_gpu_ctx *ctx, const int *tables, int n_tables)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!tables || n_tables < 1 || n_tables > GCN10_N_TABLES)
        return fail(GCN10_E_INVAL, "gcn10_gpu_set_tables: need 1..9 tables, got %d", n_tables);

    static thread_local uint8_t img16[kLut16Bytes];
    static thread_local uint8_t img1[GCN10_N_TABLES * kLut1Bytes];
    memset(img16, GCN10_NODATA, sizeof img16);
    memset(img1, GCN10_NODATA, sizeof img1);
    for (int k = 0; k < n_tables; k++) {
        for (int lc = 0; lc < 256; lc++) {
            for (int s = 0; s < 5; s++) {
                const int v = tables[(k * 256 + lc) * 5 + s];
                // src/cn.c:125-128: only values < 255 are stored, through a
                // truncating (uint8_t) cast; everything else leaves the 255
                // the raster was memset to (src/cn.c:289).
                const uint8_t b = v < GCN10_NODATA ? (uint8_t)v : (uint8_t)GCN10_NODATA;
                img16[s * kPlane16 + lc * 16 + k] = b;
                img1[k * kLut1Bytes + s * kPlane1 + lc] = b;
            }
        }
    }
    // synchronous copies: set_tables is a once-per-run call and the images are
    // reused by every later launch on any stream
    HIP_TRY(hipMemcpy(ctx->d_lut16, img16, sizeof img16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_lut1, img1, sizeof img1, hipMemcpyHostToDevice));
    ctx->n_tables = n_tables;

    // ---- pixel classes for the fused tile encoder (gcn10_deflate.hip): two pixels are in one
    // class iff all 18 rasters agree on them, i.e. iff their (soil code, landcover) pairs have
    // the same 18-vector of CN values.  Class 0 is the zero padding outside the raster.
    {
        static thread_local uint8_t class_of[gcn10::kClassCodes * 256];
        static thread_local uint8_t class_val[GCN10_N_RASTERS * 256];
        static thread_local uint8_t vecs[257][GCN10_N_RASTERS];
        int n_classes = 1;
        memset(vecs[0], 0, sizeof vecs[0]);
        memset(class_val, 0, sizeof class_val);
        bool fits = true;
        for (int cc = 0; cc < gcn10::kClassCodes && fits; cc++) {
            const int sd = cc / 6, su = cc % 6;
            for (int lc = 0; lc < 256 && fits; lc++) {
                uint8_t v[GCN10_N_RASTERS];
                for (int r = 0; r < GCN10_N_RASTERS; r++) {
                    const int sgrp = r < 9 ? sd : su;
                    const int k = r % 9;
                    v[r] = (sgrp < 5 && k < n_tables) ? img16[sgrp * kPlane16 + lc * 16 + k]
                                                      : (uint8_t)GCN10_NODATA;
                }
                int id = -1;
                for (int c = 1; c < n_classes; c++)
                    if (memcmp(vecs[c], v, sizeof v) == 0) {
                        id = c;
                        break;
                    }
                if (id < 0) {
                    if (n_classes == 256) {
                        fits = false;
                        break;
                    }
                    id = n_classes++;
                    memcpy(vecs[id], v, sizeof v);
                    for (int r = 0; r < GCN10_N_RASTERS; r++)
                        class_val[r * 256 + id] = v[r];
                }
                class_of[cc * 256 + lc] = (uint8_t)id;
            }
        }
        ctx->n_classes = fits ? n_classes : 0;      // 0: more than 256 classes, fused encoder unavailable
        if (fits) {
            if (!ctx->d_class_of)
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_class_of), sizeof class_of + sizeof class_val));
            HIP_TRY(hipMemcpy(ctx->d_class_of, class_of, sizeof class_of, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(ctx->d_class_of + sizeof class_of, class_val, sizeof class_val,
                              hipMemcpyHostToDevice));
        }
    }
    return GCN10_OK;
}

// ---- per-function kernels -------------------------------------------------

int gcn10_gpu_resample(gcn10_gpu_ctx *ctx, const uint8_t *coarse, int hsx, int hsy,
                       const int32_t *ci, const int32_t *cj, int W, int rows, uint8_t *out,
                       gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (hsx <= 0 || hsy <= 0 || W < 0 || rows < 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_resample: bad shape %dx%d <- %dx%d", W, rows, hsx, hsy);
    if (W == 0 || rows == 0)
        return GCN10_OK;
    if (!coarse || !ci || !cj || !out)
        return fail(GCN10_E_INVAL, "gcn10_gpu_resample: null pointer");
    const uint64_t npix = (uint64_t)W * rows;
    const uint32_t grid = stream_grid(ctx, (npix + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(resample_rows, dim3(grid), dim3(kThreads), 0, as_stream(ctx, stream),
                       coarse, (uint32_t)hsx, ci, cj, (uint32_t)W, (uint32_t)rows, out);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

int gcn10_gpu_modify_hysogs_data(gcn10_gpu_ctx *ctx, uint8_t *h, size_t npix, int drained,
                                 gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (npix == 0)
        return GCN10_OK;
    if (!h)
        return fail(GCN10_E_INVAL, "gcn10_gpu_modify_hysogs_data: null pointer");
    size_t head = (16 - (reinterpret_cast<uintptr_t>(h) & 15u)) & 15u;
    if (head > npix)
        head = npix;
    const size_t nvec = (npix - head) / 16;
    const uint32_t grid = stream_grid(ctx, (nvec + kThreads - 1) / kThreads + 1);
    hipLaunchKernelGGL(modify_hysogs_kernel, dim3(grid), dim3(kThreads), 0,
                       as_stream(ctx, stream), h, npix, drained, head, nvec);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

int gcn10_gpu_calculate_cn(gcn10_gpu_ctx *ctx, const uint8_t *esa, const uint8_t *hsg,
                           size_t npix, int table_index, uint8_t *out, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (ctx->n_tables == 0)
        return fail(GCN10_E_STATE, "gcn10_gpu_calculate_cn: call gcn10_gpu_set_tables first");
    if (table_index < 0 || table_index >= ctx->n_tables)
        return fail(GCN10_E_INVAL, "gcn10_gpu_calculate_cn: table %d not loaded (have %d)",
                    table_index, ctx->n_tables);
    if (npix == 0)
        return GCN10_OK;
    if (!esa || !hsg || !out)
        return fail(GCN10_E_INVAL, "gcn10_gpu_calculate_cn: null pointer");
    const uint8_t *img = ctx->d_lut1 + (size_t)table_index * kLut1Bytes;
    const bool vec = aligned16(esa) && aligned16(hsg) && aligned16(out);
    const uint64_t work = vec ? (npix / 16 + 2 * kThreads - 1) / (2 * kThreads) + 1
                              : (npix + kThreads - 1) / kThreads;
    const uint32_t grid = stream_grid(ctx, work);
    if (vec)
        hipLaunchKernelGGL(calculate_cn_kernel<true>, dim3(grid), dim3(kThreads), 0,
                           as_stream(ctx, stream), esa, hsg, npix, img, out);
    else
        hipLaunchKernelGGL(calculate_cn_kernel<false>, dim3(grid), dim3(kThreads), 0,
                           as_stream(ctx, stream), esa, hsg, npix, img, out);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

// ---- fused block path -----------------------------------------------------

int gcn10_gpu_prepare_tile(gcn10_gpu_ctx *ctx, const uint8_t *coarse, int hsx, int hsy,
                           const int32_t *ci, int W, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (hsx <= 0 || hsy <= 0 || W <= 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_prepare_tile: bad shape W=%d soil=%dx%d", W, hsx, hsy);
    if (!coarse || !ci)
        return fail(GCN10_E_INVAL, "gcn10_gpu_prepare_tile: null pointer");
    // rows padded to a multiple of 16 plus one extra group so that any 16-byte
    // read starting below W stays inside the row
    const uint32_t stride = (((uint32_t)W + 15u) & ~15u) + 16u;
    const size_t need = (size_t)stride * (size_t)hsy;
    if (need > ctx->hx_capacity) {
        // growing the workspace is the one allocation on this path; it happens
        // once per run for equally sized blocks (sync: the old buffer may be in use).
        // 16 bytes in front of row 0: the strip kernel's second soil load of a lane that
        // straddles a row end starts up to 15 bytes before a soil row.
        HIP_TRY(hipDeviceSynchronize());
        if (ctx->d_hx_alloc)
            HIP_TRY(hipFree(ctx->d_hx_alloc));
        ctx->d_hx_alloc = ctx->d_hx = nullptr;
        ctx->d_hx4 = ctx->d_hx4_complex = nullptr;
        ctx->hx4_ready = false;
        ctx->hx_capacity = 0;
        // behind the code bytes: the compact words (a quarter of the bytes) and their "complex" flag
        const size_t need16 = (need + 15) & ~(size_t)15;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_hx_alloc), 16 + need16 + need16 / 4 + 16));
        ctx->d_hx = ctx->d_hx_alloc + 16;
        ctx->d_hx4 = reinterpret_cast<uint32_t *>(ctx->d_hx + need16);
        ctx->d_hx4_complex = reinterpret_cast<uint32_t *>(ctx->d_hx + need16 + need16 / 4);
        // on the launch's own stream: ordered before expand_x_codes whatever the blocking mode of the stream
        HIP_TRY(hipMemsetAsync(ctx->d_hx4_complex, 0, 4, as_stream(ctx, stream)));
        ctx->hx4_gen = 0;
        ctx->hx_capacity = need;
    }
    uint32_t *hx4 = ctx->compact_soil ? ctx->d_hx4 : nullptr;
    ctx->hx4_ready = hx4 != nullptr;
    if (hx4 && ++ctx->hx4_gen == 0u)
        ctx->hx4_gen = 1u;      // 0 is what the word holds before the first complex tile
    ctx->hx_stride = stride;
    ctx->hx_W = (uint32_t)W;
    ctx->hx_rows = (uint32_t)hsy;
    dim3 grid((stride / 16 + kThreads - 1) / kThreads, ((uint32_t)hsy + kExpandRows - 1) / kExpandRows);
    if (aligned16(ci))
        hipLaunchKernelGGL(expand_x_codes<true>, grid, dim3(kThreads), 0, as_stream(ctx, stream),
                           coarse, (uint32_t)hsx, (uint32_t)hsy, ci, (uint32_t)W, ctx->d_hx, stride, hx4,
                           ctx->d_hx4_complex, ctx->hx4_gen);
    else
        hipLaunchKernelGGL(expand_x_codes<false>, grid, dim3(kThreads), 0, as_stream(ctx, stream),
                           coarse, (uint32_t)hsx, (uint32_t)hsy, ci, (uint32_t)W, ctx->d_hx, stride, hx4,
                           ctx->d_hx4_complex, ctx->hx4_gen);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

int gcn10_gpu_cn_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows,
                       const int32_t *cj, unsigned cond_mask, unsigned table_mask,
                       uint8_t *const out[GCN10_N_RASTERS], gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (ctx->n_tables == 0)
        return fail(GCN10_E_STATE, "gcn10_gpu_cn_strip: call gcn10_gpu_set_tables first");
    if (!ctx->d_hx || ctx->hx_W == 0)
        return fail(GCN10_E_STATE, "gcn10_gpu_cn_strip: call gcn10_gpu_prepare_tile first");
    if (W <= 0 || rows < 0 || (uint32_t)W != ctx->hx_W)
        return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: W=%d rows=%d does not match prepared tile W=%u",
                    W, rows, ctx->hx_W);
    if ((uint64_t)W * (uint64_t)rows > 0x7fffffffull)
        return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: strip of %d x %d exceeds 2^31-1 pixels", W, rows);
    if (cond_mask == 0 || (cond_mask & ~3u))
        return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: cond_mask 0x%x", cond_mask);
    if (table_mask == 0 || (table_mask >> ctx->n_tables))
        return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: table_mask 0x%x with %d tables loaded",
                    table_mask, ctx->n_tables);
    if (rows == 0)
        return GCN10_OK;
    if (!esa || !cj || !out)
        return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: null pointer");

    StripParams p;
    memset(&p, 0, sizeof p);
    bool all_aligned = aligned16(esa);
    for (int c = 0; c < 2; c++) {
        for (int k = 0; k < 9; k++) {
            const int r = c * 9 + k;
            if ((cond_mask & (1u << c)) && (table_mask & (1u << k))) {
                if (!out[r])
                    return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: out[%d] is null but selected", r);
                p.out[r] = out[r];
                all_aligned = all_aligned && aligned16(out[r]);
            }
        }
    }
    p.esa = esa;
    p.hx = ctx->d_hx;
    p.cj = cj;
    p.W = (uint32_t)W;
    p.rows = (uint32_t)rows;
    p.npix = (uint32_t)((uint64_t)W * (uint64_t)rows);
    p.hx_stride = ctx->hx_stride;
    p.hx_rows = ctx->hx_rows;
    p.nchunks = (p.npix + kChunk - 1) / kChunk;
    p.table_mask = table_mask;
    hipStream_t s = as_stream(ctx, stream);

    p.xcd_slabs = (uint32_t)ctx->xcd_slabs;
    p.nvec16 = p.npix & ~15u;
    if (ctx->hx4_ready && ctx->compact_soil && (p.W & 15u) == 0u) {
        // rows are 16-byte aligned: a lane's 16 pixels are one column group of the compact soil words
        p.hx4 = ctx->d_hx4;
        p.hx4_complex = ctx->d_hx4_complex;
        p.hx4_stride = ctx->hx_stride / 16u;
        p.hx4_gen = ctx->hx4_gen;
    }
    // the vector kernel wants a wave's 1024-px span to cross at most one row end
    if (p.npix < 16u || p.W < kMinVectorW)
        all_aligned = false;
    if (!all_aligned) {
        p.lut = ctx->d_lut16;
        const uint32_t g = stream_grid(ctx, ((uint64_t)p.npix + kThreads - 1) / kThreads);
        if (ctx->time_start && ctx->time_stop) {
            void *args[] = { &p, &cond_mask };
            HIP_TRY(hipExtLaunchKernel(reinterpret_cast<const void *>(cn_strip_bytes), dim3(g), dim3(kThreads), args, 0,
                                       s, ctx->time_start, ctx->time_stop, 0));
            ctx->time_start = ctx->time_stop = nullptr;
        }
        else {
            hipLaunchKernelGGL(cn_strip_bytes, dim3(g), dim3(kThreads), 0, s, p, cond_mask);
        }
        ctx->last_kernel = "cn_strip_bytes";
    }
    else {
        const bool single = popcount(table_mask) == 1;
        const bool all = table_mask == 0x1ffu;
        // 0 = by the number of store streams: two chunks per trip up to nine rasters, one beyond
        const int n_streams = popcount(cond_mask) * popcount(table_mask);
        const bool tuned = single && ctx->single.set;
        const int ilp = tuned ? ctx->single.ilp
                              : single ? ctx->ilp1 : (ctx->ilp16 > 0 ? ctx->ilp16 : (n_streams > 9 ? 1 : 2));
        const bool nt = ctx->nontemporal != 0;
        if (single) {
            p.single_k = (uint32_t)__builtin_ctz(table_mask);
            p.lut = ctx->d_lut1 + (size_t)p.single_k * kLut1Bytes;
        }
        else {
            p.lut = ctx->d_lut16;
        }
        const bool pf = (tuned ? ctx->single.prefetch : ctx->prefetch) != 0 && ilp <= 2;
        p.nchunks = (p.nchunks + (uint32_t)ilp - 1) / (uint32_t)ilp;    // groups of ILP sub-chunks
        const uint32_t grid = stream_grid(ctx, p.nchunks, tuned ? ctx->single.grid_blocks_per_cu : 0);
        if (tuned)
            p.xcd_slabs = (uint32_t)ctx->single.xcd_slabs;
        strip_kernel_t fn = pick_strip_kernel(single, cond_mask, all, ilp, nt, pf);
        if (!fn)
            return fail(GCN10_E_INVAL, "gcn10_gpu_cn_strip: no kernel for ilp=%d", ilp);
        if (ctx->time_start && ctx->time_stop) {
            // events attached to the dispatch itself: they bracket the kernel, not the
            // command-processor gaps around separately recorded events
            void *args[] = { &p };
            HIP_TRY(hipExtLaunchKernel(reinterpret_cast<const void *>(fn), dim3(grid), dim3(kThreads), args, 0,
                                       s, ctx->time_start, ctx->time_stop, 0));
            ctx->time_start = ctx->time_stop = nullptr;
        }
        else {
            hipLaunchKernelGGL(fn, dim3(grid), dim3(kThreads), 0, s, p);
        }
        // the instantiation's name as rocprofv3 prints it: <KIND, COND_MASK, ALL_TABLES, ILP, NT, PF>
        snprintf(ctx->kernel_name, sizeof ctx->kernel_name, "cn_strip_kernel<%d, %u, %s, %d, %s, %s>",
                 single ? 1 : 0, cond_mask, (all || single) ? "true" : "false", ilp, nt ? "true" : "false",
                 pf ? "true" : "false");
        ctx->last_kernel = ctx->kernel_name;
    }
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

size_t gcn10_gpu_strip_algorithmic_bytes(int W, int rows, int hsx, int hsy, unsigned cond_mask,
                                         unsigned table_mask)
{
    if (W <= 0 || rows <= 0)
        return 0;
    const size_t n_out = (size_t)popcount(cond_mask & 3u) * (size_t)popcount(table_mask & 0x1ffu);
    // 1 B landcover read + n_out B written per pixel,
    // + the coarse soil window once, + the int32 index maps once.
    return (size_t)W * rows * (1 + n_out) + (size_t)(hsx > 0 ? hsx : 0) * (hsy > 0 ? hsy : 0) +
           4 * ((size_t)W + rows);
}

int gcn10_gpu_set_option(gcn10_gpu_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name)
        return fail(GCN10_E_INVAL, "gcn10_gpu_set_option: null argument");
    if (!strcmp(name, "defaults")) {
        const gcn10_gpu_ctx fresh;
        ctx->grid_blocks_per_cu = fresh.grid_blocks_per_cu;
        ctx->ilp16 = fresh.ilp16;
        ctx->ilp1 = fresh.ilp1;
        ctx->nontemporal = fresh.nontemporal;
        ctx->xcd_slabs = fresh.xcd_slabs;
        ctx->prefetch = fresh.prefetch;
        ctx->compact_soil = fresh.compact_soil;
        ctx->deflate_wave_codes = fresh.deflate_wave_codes;
        ctx->arena_segment_align = fresh.arena_segment_align;
        ctx->fused_diag = fresh.fused_diag;
        ctx->fused_parse = fresh.fused_parse;
        ctx->fused_emit = fresh.fused_emit;
        ctx->fused_stats_stop = fresh.fused_stats_stop;
        ctx->codes_stop = fresh.codes_stop;
        ctx->inflate_diag = fresh.inflate_diag;
        ctx->event_sync_sleep_us = fresh.event_sync_sleep_us;
        ctx->single = fresh.single;
    }
    else if (!strcmp(name, "grid_blocks_per_cu") && value >= 1 && value <= 64)
        ctx->grid_blocks_per_cu = value;
    else if (!strcmp(name, "ilp16") && (value == 0 || value == 1 || value == 2))
        ctx->ilp16 = value;
    else if (!strcmp(name, "ilp1") && (value == 1 || value == 2 || value == 4)) {
        ctx->ilp1 = value;
        ctx->single.set = false;        // an explicit setting replaces a calibrated shape
    }
    else if (!strcmp(name, "nontemporal") && (value == 0 || value == 1))
        ctx->nontemporal = value;
    else if (!strcmp(name, "xcd_slabs") && (value == 0 || value == 1))
        ctx->xcd_slabs = value;
    else if (!strcmp(name, "deflate_wave_codes") && (value == 0 || value == 1))
        ctx->deflate_wave_codes = value;
    else if (!strcmp(name, "arena_segment_align") && value >= 16 && value <= 4096 && (value & (value - 1)) == 0)
        ctx->arena_segment_align = value;
    else if (!strcmp(name, "codes_stop") && value >= 0 && value <= 8)
        ctx->codes_stop = value;
    else if (!strcmp(name, "fused_stats_stop") && value >= 0 && value <= 8)
        ctx->fused_stats_stop = value;
    else if (!strcmp(name, "fused_emit") && (value == 0 || value == 1))
        ctx->fused_emit = value;
    else if (!strcmp(name, "fused_parse") && (value == 0 || value == 1))
        ctx->fused_parse = value;
    else if (!strcmp(name, "fused_diag") && value >= 0 && value < 64)
        ctx->fused_diag = value;
    else if (!strcmp(name, "event_sync_sleep_us") && value >= 0 && value <= 100000)
        ctx->event_sync_sleep_us = value;
    else if (!strcmp(name, "inflate_diag") && value >= 0 && value < 8)
        ctx->inflate_diag = value;
    else if (!strcmp(name, "prefetch") && (value == -1 || value == 0 || value == 1))
        ctx->prefetch = value;
    else if (!strcmp(name, "compact_soil") && (value == 0 || value == 1))
        ctx->compact_soil = value;      // strips launched from now on; 1 needs a gcn10_gpu_prepare_tile made with it on
    else
        return fail(GCN10_E_INVAL, "gcn10_gpu_set_option: unknown option or bad value: %s=%d", name, value);
    return GCN10_OK;
}

int gcn10_gpu_time_next_strip(gcn10_gpu_ctx *ctx, gcn10_event_t start, gcn10_event_t stop)
{
    if (!ctx || !start || !stop)
        return fail(GCN10_E_INVAL, "gcn10_gpu_time_next_strip: null argument");
    ctx->time_start = reinterpret_cast<hipEvent_t>(start);
    ctx->time_stop = reinterpret_cast<hipEvent_t>(stop);
    return GCN10_OK;
}

int gcn10_gpu_stream_copy(gcn10_gpu_ctx *ctx, const void *src, void *dst, size_t bytes, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (bytes == 0)
        return GCN10_OK;
    if (!src || !dst || (bytes & 15u) || !aligned16(src) || !aligned16(dst) || bytes / 16 > 0x7fffffffull)
        return fail(GCN10_E_INVAL, "gcn10_gpu_stream_copy: needs 16-byte aligned pointers and size (< 32 GiB)");
    const u32x4 *in = static_cast<const u32x4 *>(src);
    u32x4 *out = static_cast<u32x4 *>(dst);
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2u * kThreads - 1u) / (2u * kThreads);
    const uint32_t grid = stream_grid(ctx, ntrips);
    hipStream_t s = as_stream(ctx, stream);
    if (ctx->time_start && ctx->time_stop) {
        void *args[] = { &in, &out, &nvec, &ntrips };
        HIP_TRY(hipExtLaunchKernel(reinterpret_cast<const void *>(stream_copy_kernel), dim3(grid), dim3(kThreads),
                                   args, 0, s, ctx->time_start, ctx->time_stop, 0));
        ctx->time_start = ctx->time_stop = nullptr;
    }
    else {
        hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(kThreads), 0, s, in, out, nvec, ntrips);
    }
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

int gcn10_gpu_tune_single_raster(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows, const int32_t *cj,
                                 unsigned cond_mask, unsigned table_mask, uint8_t *arena, size_t arena_bytes,
                                 size_t step, uint8_t **best_out, float *best_ms, char *report, size_t report_cap,
                                 gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!best_out || !arena || !esa || !cj)
        return fail(GCN10_E_INVAL, "gcn10_gpu_tune_single_raster: null pointer");
    if (popcount(cond_mask & 3u) != 1 || popcount(table_mask & 0x1ffu) != 1 || (cond_mask & ~3u) || (table_mask >> 9))
        return fail(GCN10_E_INVAL, "gcn10_gpu_tune_single_raster: exactly one condition and one table");
    if (W <= 0 || rows <= 0 || (uint64_t)W * (uint64_t)rows > 0x7fffffffull)
        return fail(GCN10_E_INVAL, "gcn10_gpu_tune_single_raster: bad shape %d x %d", W, rows);
    const size_t npix = (size_t)W * (size_t)rows;
    if (step == 0 || (step & 255u) || !aligned16(arena) || arena_bytes < npix)
        return fail(GCN10_E_INVAL, "gcn10_gpu_tune_single_raster: arena of %zu bytes / step %zu cannot hold a raster of %zu",
                    arena_bytes, step, npix);
    const int r = (cond_mask & 2u ? 9 : 0) + __builtin_ctz(table_mask);
    struct Shape { int xcd, bpc, ilp, pf; };
    // XCD slabs only: a grid-stride sweep is 1-2 % faster at its best positions, but every XCD then pulls
    // every soil row through the fabric (FETCH_SIZE 1.18x the algorithmic bytes against 1.03x, profiles/r02)
    static const Shape shapes[] = { {1, 8, 2, 1}, {1, 16, 2, 1}, {1, 8, 4, 0}, {1, 16, 4, 0}, {1, 16, 1, 0}, {1, 8, 2, 0},
                                    {1, 16, 2, 0}, {1, 8, 1, 1} };
    const gcn10_gpu_ctx::SingleShape saved = ctx->single;
    hipStream_t s = as_stream(ctx, stream);
    constexpr int kRuns = 3;
    hipEvent_t e0[kRuns], e1[kRuns];
    for (int i = 0; i < kRuns; i++) {
        e0[i] = e1[i] = nullptr;
        if (hipEventCreate(&e0[i]) != hipSuccess || hipEventCreate(&e1[i]) != hipSuccess)
            return fail(GCN10_E_HIP, "gcn10_gpu_tune_single_raster: event creation failed");
    }
    float best = 1e30f, worst = 0.f;
    size_t best_pos = 0;
    Shape best_shape = shapes[0];
    int n_pos = 0;
    rc = GCN10_OK;
    for (size_t pos = 0; pos + npix <= arena_bytes && rc == GCN10_OK; pos += step, n_pos++) {
        uint8_t *out[GCN10_N_RASTERS] = { nullptr };
        out[r] = arena + pos;
        for (const Shape &sh : shapes) {
            ctx->single.set = true;
            ctx->single.xcd_slabs = sh.xcd;
            ctx->single.grid_blocks_per_cu = sh.bpc;
            ctx->single.ilp = sh.ilp;
            ctx->single.prefetch = sh.pf;
            rc = gcn10_gpu_cn_strip(ctx, esa, W, rows, cj, cond_mask, table_mask, out, stream);     // untimed
            for (int i = 0; i < kRuns && rc == GCN10_OK; i++) {
                ctx->time_start = e0[i];
                ctx->time_stop = e1[i];
                rc = gcn10_gpu_cn_strip(ctx, esa, W, rows, cj, cond_mask, table_mask, out, stream);
            }
            if (rc != GCN10_OK)
                break;
            if (hipStreamSynchronize(s) != hipSuccess) {
                rc = fail(GCN10_E_HIP, "gcn10_gpu_tune_single_raster: stream synchronisation failed");
                break;
            }
            float ms[kRuns];
            for (int i = 0; i < kRuns; i++)
                if (hipEventElapsedTime(&ms[i], e0[i], e1[i]) != hipSuccess)
                    ms[i] = 1e30f;
            // median of three
            const float lo = ms[0] < ms[1] ? ms[0] : ms[1], hi = ms[0] < ms[1] ? ms[1] : ms[0];
            const float med = ms[2] < lo ? lo : (ms[2] > hi ? hi : ms[2]);
            if (med < best) {
                best = med;
                best_pos = pos;
                best_shape = sh;
            }
            if (med > worst && med < 1e29f)
                worst = med;
        }
    }
    for (int i = 0; i < kRuns; i++) {
        (void)hipEventDestroy(e0[i]);
        (void)hipEventDestroy(e1[i]);
    }
    ctx->time_start = ctx->time_stop = nullptr;
    ctx->single = saved;
    if (rc != GCN10_OK)
        return rc;
    ctx->single.set = true;
    ctx->single.xcd_slabs = best_shape.xcd;
    ctx->single.grid_blocks_per_cu = best_shape.bpc;
    ctx->single.ilp = best_shape.ilp;
    ctx->single.prefetch = best_shape.pf;
    *best_out = arena + best_pos;
    if (best_ms)
        *best_ms = best;
    if (report && report_cap)
        snprintf(report, report_cap,
                 "{\"positions\": %d, \"step_bytes\": %zu, \"shapes\": %zu, \"best_ms\": %.4f, \"worst_ms\": %.4f, "
                 "\"best_offset_bytes\": %zu, \"xcd_slabs\": %d, \"grid_blocks_per_cu\": %d, \"ilp1\": %d, \"prefetch\": %d}",
                 n_pos, step, sizeof shapes / sizeof shapes[0], best, worst, best_pos, best_shape.xcd, best_shape.bpc,
                 best_shape.ilp, best_shape.pf);
    return GCN10_OK;
}

int gcn10_gpu_soil_words_state(gcn10_gpu_ctx *ctx, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (!ctx->hx4_ready || !ctx->d_hx4_complex)
        return 0;
    uint32_t flag = 0;
    hipStream_t s = as_stream(ctx, stream);
    HIP_TRY(hipMemcpyAsync(&flag, ctx->d_hx4_complex, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return flag == ctx->hx4_gen ? 2 : 1;
}

const char *gcn10_gpu_last_kernel_name(gcn10_gpu_ctx *ctx)
{
    return ctx ? ctx->last_kernel : "";
}

}  // extern "C"

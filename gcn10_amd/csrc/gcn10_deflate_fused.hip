// gcn10_deflate_fused.hip -- the fused tile encoder: 18 zlib streams per 256x256 tile position
// straight from the landcover strip and the prepared soil codes; no CN raster in HBM.
// What it replaces and the stream format: see gcn10_deflate.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "gcn10_deflate_internal.hpp"
#include "gcn10_gpu.h"
#include "gcn10_gpu_internal.hpp"

using gcn10::as_stream;
using gcn10::fail;
using gcn10::u32x4;
using gcn10::use_device;
using namespace gcn10_deflate;

namespace {

// ------------------------------------------------------------------------
// Fused tile encoder: the 18 rasters of a block are functions of the same (landcover,
// soil) pixel pairs, so their repeats line up.  Pixels are reduced to a CLASS id (two
// pixels share a class iff all 18 rasters agree on them; gcn10_gpu_set_tables builds
// the map), the class tile is loaded, masked and tokenised ONCE per tile position, and
// the 18 streams are produced from it: no CN raster is ever written to or read from HBM
// (24.6 GB written + 2 x 23.3 GB read per block in the unfused path -> 1.4 GB read).
// A match of the class stream is a match in every raster; a raster's own extra repeats
// (two classes with the same value in that raster) are coded as literals, which costs a
// little compression.  Three launches per strip:
//   F-A  one workgroup per tile position: class tile, match masks, greedy parse of every
//        row, the position's 16-bit token stream -> workspace; per raster the literal
//        statistics (class counts mapped through class_val) and the Adler-32 (per-class
//        pixel counts and position-weight sums); drained/undrained aliases
//   B    code construction per (raster, tile), the per-raster encoder's pass unchanged
//   F-C  one workgroup per (position, group of kGroup rasters), token-parallel: prefix sums
//        of the tokens' code lengths place their bits, words go straight to the arena
// ------------------------------------------------------------------------
constexpr int kGroup = 6;            // rasters emitted by one workgroup of pass F-C

struct FusedJob {
    const uint8_t *esa;
    const uint8_t *hx;
    const int32_t *cj;
    const uint8_t *class_of;        // [36][256]; class_val [18][256] follows
    uint16_t *tok;                  // [positions][kTokStride] token stream of each tile position (F-A -> F-C)
    uint32_t *n_tok;                // [positions] its length, end-of-block token included
    uint32_t hx_stride, hx_rows;
    uint32_t diag;                  // gcn10_gpu_set_option("fused_diag"): timing experiments
    uint32_t seg_align;             // every raster's extent of the strip starts at a multiple of this (option arena_segment_align)
    uint32_t n_sel;                 // selected rasters, ascending
    uint8_t sel[GCN10_N_RASTERS];
    TileJob t;
};

__device__ __forceinline__ uint32_t compact_code(uint32_t code)
{
    return (code & 15u) * 6u + (code >> 4);
}

// Class ids of pixels (x..x+3, y) of the strip, one per byte; class 0 outside the raster.
__device__ __forceinline__ uint32_t class_pixels4(const FusedJob &job, uint32_t x, uint32_t y,
                                                  const uint8_t *class_of_lds)
{
    typedef uint32_t u32_u __attribute__((aligned(1)));
    const uint32_t W = job.t.W;
    uint32_t out = 0;
    if (y < job.t.rows && x < W) {
        uint32_t srow = (uint32_t)job.cj[y];
        srow = srow < job.hx_rows ? srow : job.hx_rows - 1u;
        const uint8_t *pe = job.esa + (size_t)y * W + x;
        // hx rows are padded by >= 16 bytes past W: a 4-byte read starting below W is safe
        const uint32_t c4 = *reinterpret_cast<const u32_u *>(job.hx + (size_t)srow * job.hx_stride + x);
        uint32_t e4 = 0;
        if (x + 4u <= W) {
            e4 = *reinterpret_cast<const u32_u *>(pe);
        }
        else {
            for (uint32_t k = 0; x + k < W; k++)
                e4 |= (uint32_t)pe[k] << (8 * k);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            if (x + q < W) {
                const uint32_t lc = (e4 >> (8 * q)) & 0xffu;
                const uint32_t cc = compact_code((c4 >> (8 * q)) & 0xffu);
                out |= (uint32_t)class_of_lds[cc * 256u + lc] << (8 * q);
            }
        }
    }
    return out;
}

// Class ids of one tile position into LDS (row stride kRowStride).  `soil_row` holds the soil
// row cj[y] of the tile's 256 rows (LDS), so that the landcover and soil-code loads of a tile
// depend on nothing and go out 16 rows at a time.
__device__ __forceinline__ void load_class_tile(const FusedJob &job, uint32_t tx, uint32_t ty,
                                                const uint8_t *class_of_lds, const uint32_t *soil_row,
                                                uint8_t *tile, int t)
{
    typedef uint32_t u32_u __attribute__((aligned(1)));
    const uint32_t x = tx * kTile + (uint32_t)(t & 63) * 4u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(tile) + (t & 63);
    if ((tx + 1) * kTile <= job.t.W && (ty + 1) * kTile <= job.t.rows) {
        // interior tile: no edge cases
        const uint8_t *pe = job.esa + ((size_t)ty * kTile + (uint32_t)(t >> 6)) * job.t.W + x;
        const uint8_t *ph = job.hx + x;
        for (int b = 0; b < kTile / 4; b += 16) {
            uint32_t e4[16], c4[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = (b + i) * 4 + (t >> 6);
                e4[i] = *reinterpret_cast<const u32_u *>(pe + (size_t)(b + i) * 4 * job.t.W);
                c4[i] = *reinterpret_cast<const u32_u *>(ph + (size_t)soil_row[r] * job.hx_stride);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = (b + i) * 4 + (t >> 6);
                uint32_t out = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t lc = (e4[i] >> (8 * q)) & 0xffu;
                    const uint32_t cc = compact_code((c4[i] >> (8 * q)) & 0xffu);
                    out |= (uint32_t)class_of_lds[cc * 256u + lc] << (8 * q);
                }
                dst[r * (kRowStride / 4)] = out;
            }
        }
        return;
    }
#pragma unroll 4
    for (int i = 0; i < kTile / 4; i++) {
        const int r = i * 4 + (t >> 6);
        dst[r * (kRowStride / 4)] = class_pixels4(job, x, ty * kTile + (uint32_t)r, class_of_lds);
    }
}

__device__ __forceinline__ void set_bit(unsigned long long (&m)[4], int x)
{
    const unsigned long long b = 1ull << (x & 63);
    const int w = x >> 6;
    m[0] |= w == 0 ? b : 0ull;
    m[1] |= w == 1 ? b : 0ull;
    m[2] |= w == 2 ? b : 0ull;
    m[3] |= w == 3 ? b : 0ull;
}

// next position >= x with a set bit (256 if none)
__device__ __forceinline__ int next_set(const unsigned long long (&m)[4], int x)
{
    for (int pos = x; pos < kTile;) {
        const int b = pos & 63;
        const unsigned long long v = pick(m, pos >> 6) >> b;
        if (v)
            return pos + __builtin_ctzll(v);
        pos += 64 - b;
    }
    return kTile;
}

// The greedy parse of parse_row(), done once per tile position and kept: statistics are
// counted, and the row is rewritten IN PLACE as its token stream.  A literal stays the byte
// it was (a class id); a match of length len at x (it covers >= 3 bytes) becomes
//   row[x]   = length code (0..28) | 0x80 if its distance is 256
//   row[x+1] = value of the length's extra bits | number of extra bits << 5
//   row[x+2] = len - 3
// and bit x of `start` is set.  Every mask of the tile must have been computed before.
// Returns the number of tokens of the row.
__device__ __forceinline__ uint32_t tokenise_row(uint8_t *tile, int t, const RowMasks &m, uint32_t *lit_hist,
                                                 uint32_t *dist_hist, unsigned long long (&start)[4], bool diag_no_lit)
{
    uint8_t *row = tile + t * kRowStride;
    int x = 0;
    uint32_t n_tokens = 0;
    start[0] = start[1] = start[2] = start[3] = 0ull;
    while (x < kTile) {
        const int cand = next_candidate(m, x);
        n_tokens += (uint32_t)(cand - x);
        if (diag_no_lit)
            x = cand;
        for (; x < cand; x++)
            atomicAdd(&lit_hist[row[x]], 1u);
        if (x >= kTile)
            break;
        n_tokens++;
        const int l1 = run_from(m.near_, x);
        const int l256 = run_from(m.far_, x);
        const bool far = l256 > l1;                 // tie: distance 1 (no extra bits)
        const int len = far ? l256 : l1;
        if (len >= 3) {
            const int lc = length_code(len);
            const int ne = (lc < 8 || lc == 28) ? 0 : (lc - 4) >> 2;
            atomicAdd(&lit_hist[257 + lc], 1u);
            atomicAdd(&dist_hist[far ? 1 : 0], 1u);
            row[x] = (uint8_t)(lc | (far ? 0x80 : 0));
            row[x + 1] = (uint8_t)((len - kLenBase[lc]) | (ne << 5));
            row[x + 2] = (uint8_t)(len - 3);
            set_bit(start, x);
            x += len;
        }
        else {
            atomicAdd(&lit_hist[row[x]], 1u);
            x++;
        }
    }
    return n_tokens;
}

// The token stream pass F-C reads, 16 bits per token, rows back to back:
//   0x0000 | class                                  literal
//   0x8000 | length code | extra value << 5 | far << 10   match (far: distance 256, else 1)
//   0x4000                                          end of block
constexpr uint32_t kTokMatch = 0x8000u, kTokEnd = 0x4000u;
constexpr uint32_t kTokStride = kTileBytes + 64;    // tokens per tile position, worst case + the end token

__device__ __forceinline__ uint32_t wave_sum64(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

struct SharedFA {
    uint8_t tile[kTile * kRowStride];
    uint32_t soil_row[kTile];           // cj[y] of the tile's rows, clamped
    union {
        uint8_t class_of[gcn10::kClassCodes * 256];     // while the tile is built
        struct {
            uint32_t lit_hist[288];     // literals by class, match length symbols at 257..
            uint32_t dist_hist[2];
            uint32_t n_c[256], w_c[256];                // pixels per class, sum of their Adler weights
            uint32_t H[kGroup][256];                    // literal counts by VALUE, kGroup rasters at a time
            uint32_t s1[GCN10_N_RASTERS], s2[GCN10_N_RASTERS];
            uint32_t row_base[4];
            uint32_t sig[GCN10_N_RASTERS], cand[GCN10_N_RASTERS], differ;
        } a;
    };
};

// pass F-A: one workgroup per tile position.  Classes -> tokens + statistics of the class
// stream -> per raster: value statistics and Adler-32 (from per-class pixel counts and
// position weights, no raster byte is ever formed).
__global__ __launch_bounds__(kTile) void fused_stats_kernel(const FusedJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedFA &sh = *reinterpret_cast<SharedFA *>(smem);
    const int t = threadIdx.x;
    const uint32_t tiles = job.t.across * job.t.down;
    const uint32_t tix = blockIdx.x;
    const uint32_t ty = tix / job.t.across, tx = tix - ty * job.t.across;
    const uint8_t *class_val = job.class_of + gcn10::kClassCodes * 256;
    // (pass B adds the stream sizes up per raster and 64 positions for pass F-C's placement: from zero)
    if (job.t.chunk_tot && tix == 0u)
        for (uint32_t i = (uint32_t)t; i < job.n_sel * job.t.n_chunks; i += blockDim.x)
            job.t.chunk_tot[i] = 0u;

    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kTile)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    {
        const uint32_t y = ty * kTile + (uint32_t)t;
        uint32_t srow = y < job.t.rows ? (uint32_t)job.cj[y] : 0u;
        sh.soil_row[t] = srow < job.hx_rows ? srow : job.hx_rows - 1u;
    }
    __syncthreads();
    load_class_tile(job, tx, ty, sh.class_of, sh.soil_row, sh.tile, t);
    __syncthreads();
    for (int i = t; i < 288; i += kTile)
        sh.a.lit_hist[i] = 0;
    if (t < 2)
        sh.a.dist_hist[t] = 0;
    sh.a.n_c[t] = 0;
    sh.a.w_c[t] = 0;
    if (t < GCN10_N_RASTERS) {
        sh.a.s1[t] = 0;
        sh.a.s2[t] = 0;
    }
    RowMasks m;
    row_masks(sh.tile, t, m);
    __syncthreads();                                // every mask is built: rows may be rewritten

    // pixels and Adler weights per class, run by run along row t (weight of byte i of the tile:
    // 65536 - i).  Runs end where the "equals its left neighbour" mask has a zero, so a row costs
    // as many steps as it has runs, not 256.
    {
        const uint8_t *row = sh.tile + t * kRowStride;
        unsigned long long brk[4] = { ~m.near_[0] | 1ull, ~m.near_[1], ~m.near_[2], ~m.near_[3] };
        const uint32_t w0 = (uint32_t)kTileBytes - (uint32_t)t * kTile;        // weight of the row's first byte
        int x0 = 0;
        while (x0 < kTile) {
            const int x1 = next_set(brk, x0 + 1);           // the next run's start, or the row's end
            const uint32_t n = (uint32_t)(x1 - x0);
            const uint32_t c = row[x0];
            atomicAdd(&sh.a.n_c[c], n);
            atomicAdd(&sh.a.w_c[c], n * w0 - ((n * (uint32_t)(x0 + x1 - 1)) >> 1));
            x0 = x1;
        }
    }
    unsigned long long start[4];
    const uint32_t my_tokens = tokenise_row(sh.tile, t, m, sh.a.lit_hist, sh.a.dist_hist, start, (job.diag & 8u) != 0u);
    // rows back to back: where this row's tokens go
    {
        uint32_t incl = my_tokens;
        const int lane = t & 63;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += up;
        }
        if (lane == 63)
            sh.a.row_base[t >> 6] = incl;
        __syncthreads();
        uint32_t at = incl - my_tokens;
        for (int w = 0; w < (t >> 6); w++)
            at += sh.a.row_base[w];
        uint16_t *out = job.tok + (size_t)tix * kTokStride;
        const uint8_t *row = sh.tile + t * kRowStride;
        int x = (job.diag & 16u) ? kTile : 0;       // (timing experiment: no token write-out)
        while (x < kTile) {
            const int p = next_set(start, x);
            for (; x < p; x++)
                out[at++] = (uint16_t)row[x];
            if (x >= kTile)
                break;
            const uint32_t b0 = row[x], b1 = row[x + 1], b2 = row[x + 2];
            out[at++] = (uint16_t)(kTokMatch | (b0 & 31u) | (b1 & 31u) << 5 | (b0 >> 7) << 10);
            x += (int)b2 + 3;
        }
        if (t == kTile - 1) {
            out[at] = (uint16_t)kTokEnd;
            job.n_tok[tix] = at + 1u;
        }
    }
    __syncthreads();

    // Adler-32 of every raster's tile: thread t = class t, s1 = 1 + sum n_c val, s2 = N + sum w_c val;
    // and a hash of (class, value) over the classes that occur, per raster (alias detection below)
    const uint32_t lits = sh.a.lit_hist[t];
    {
        const uint32_t n = sh.a.n_c[t];
        const uint32_t w = sh.a.w_c[t] % 65521u;
        const bool present = n != 0;
        if (t < GCN10_N_RASTERS)
            sh.a.sig[t] = 0;
        if (t == 0)
            sh.a.differ = 0;
        __syncthreads();
        for (uint32_t j = 0; j < job.n_sel; j++) {
            const uint32_t v = class_val[job.sel[j] * 256 + t];
            uint32_t h = present ? (((uint32_t)t * 0x9E3779B1u + v * 0x85EBCA6Bu + 0x27D4EB2Fu) * 0x165667B1u) : 0u;
            h ^= h >> 15;
            const uint32_t p1 = wave_sum64(n * v), p2 = wave_sum64(w * v), p3 = wave_sum64(h);
            if ((t & 63) == 0) {
                atomicAdd(&sh.a.s1[j], p1);
                atomicAdd(&sh.a.s2[j], p2);
                atomicAdd(&sh.a.sig[j], p3);
            }
        }
    }
    // Rasters that cannot differ on this tile: two selected rasters whose tables map every class
    // that OCCURS here to the same value are the same bytes on it (a table's drained and undrained
    // raster wherever no dual soil class lies under the tile; all 18 on open water, ice, no-data).
    // The later raster becomes an alias of the earliest equal one.  The hash proposes the partner,
    // a comparison class by class confirms it.
    uint32_t alias_of[GCN10_N_RASTERS];
    {
        const bool present = sh.a.n_c[t] != 0;
        __syncthreads();
        if ((uint32_t)t < job.n_sel) {
            uint32_t cand = 0xffu;
            for (uint32_t q = 0; q < (uint32_t)t; q++)
                if (cand == 0xffu && sh.a.sig[q] == sh.a.sig[t])
                    cand = q;
            sh.a.cand[t] = cand;
        }
        __syncthreads();
        {
            uint32_t mine = 0;
            if (present)
                for (uint32_t j = 1; j < job.n_sel; j++) {
                    const uint32_t q = sh.a.cand[j];
                    if (q != 0xffu && class_val[job.sel[j] * 256 + t] != class_val[job.sel[q] * 256 + t])
                        mine |= 1u << j;
                }
            if (mine)
                atomicOr(&sh.a.differ, mine);
        }
        __syncthreads();
        for (uint32_t j = 0; j < job.n_sel; j++) {
            const uint32_t q = sh.a.cand[j];
            alias_of[j] = (q != 0xffu && !((sh.a.differ >> j) & 1u)) ? (kAliasFlag | q) : 0u;
        }
    }
    // per raster: literal counts by VALUE, kGroup rasters per round
    for (uint32_t j0 = 0; j0 < job.n_sel; j0 += kGroup) {
        const uint32_t nj = job.n_sel - j0 < (uint32_t)kGroup ? job.n_sel - j0 : (uint32_t)kGroup;
        for (uint32_t k = 0; k < nj; k++)
            sh.a.H[k][t] = 0;
        __syncthreads();
        if (lits)
            for (uint32_t k = 0; k < nj; k++)
                atomicAdd(&sh.a.H[k][class_val[job.sel[j0 + k] * 256 + t]], lits);
        __syncthreads();
        for (uint32_t k = 0; k < nj; k++) {
            const uint32_t j = j0 + k;
            uint32_t *out = job.t.hist + ((size_t)j * tiles + tix) * kHistWords;
            for (int i = t; i < kHistWords; i += kTile) {
                uint32_t v;
                if (i < 256)
                    v = sh.a.H[k][i];
                else if (i == 256)
                    v = 1u;                             // end of block
                else if (i < 288)
                    v = sh.a.lit_hist[i];               // match length symbols: the same for every raster
                else if (i < 290)
                    v = sh.a.dist_hist[i - 288];
                else if (i == 290)
                    v = (((65536u % 65521u + sh.a.s2[j] % 65521u) % 65521u) << 16) |
                        ((1u + sh.a.s1[j]) % 65521u);
                else if (i == 291)
                    v = alias_of[j];
                else
                    v = 0u;
                out[i] = v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------
// pass F-A, segment-parallel form (round 3, the default).  Same outputs as fused_stats_kernel
// -- the SAME token stream per tile position, bit for bit, and the same statistics of every raster --
// with four times the lanes: 1024 threads per tile position, ONE LANE PER 64-PIXEL SEGMENT of a
// tile row, so a lane's two match-candidate masks are one 64-bit word each and its serial walk is at
// most 64 pixels long (the row-per-lane form walks 256 pixels with four-word masks; its workgroup of
// four waves, two workgroups per CU by LDS, leaves the SIMDs two waves each -- SQ counters of round
// 2: a wave issues 42 % of the time and waits 50 %).  With 16 waves per workgroup the same 66.5 KB of
// LDS feed eight waves per SIMD.
// The greedy parse of a row is a chain (a match decides where the next token starts), and matches
// cross segment boundaries.  It is kept exact in two steps.  (1) Every lane publishes how many
// leading pixels of its segment continue a run from the segment before ("leads", for both
// distances), so a lane can tell the full length of a match that leaves its segment, and every lane
// parses its segment SPECULATIVELY from pixel 0.  (2) Segment by segment (three workgroup barriers),
// a lane reads where the last match of the segment before it really ended -- its entry offset --
// and, if that is not 0, re-parses from there only until it stands on a token start of its
// speculative parse: greedy parsing depends on the position alone, so from that pixel on the two
// parses are the same and the speculative tokens are kept (typically after one or two tokens).
// Tokens are written in raster order (row-major, segment by segment), so pass F-C, pass B and the
// host see exactly what the row form produced.
// ------------------------------------------------------------------------
constexpr int kSegs = 4;
constexpr int kSegPx = kTile / kSegs;               // 64
constexpr int kFA2Threads = kTile * kSegs;          // 1024

struct SharedFA2 {
    uint8_t tile[kTile * kRowStride];
    uint32_t soil_row[kTile];
    union {
        uint8_t class_of[gcn10::kClassCodes * 256];     // while the tile is built
        struct {
            uint32_t lit_hist[288];
            uint32_t dist_hist[2];
            uint32_t n_c[2][256], w_c[2][256];  // two copies, by lane parity: half the lanes per LDS atomic on one class
            union {
                uint32_t H[kGroup][256];                    // after the tokens are out
                struct {
                    uint32_t seg_at[kFA2Threads];           // tokens of (row, seg) in raster order, then their exclusive prefix
                    uint32_t wave_tot[kFA2Threads / 64];
                    uint8_t lead_n[kTile][kSegs];           // leading pixels of (row, seg) that continue a distance-1 run
                    uint8_t lead_f[kTile][kSegs];           // ... that equal the row above (0 .. 64)
                    uint16_t exit_at[kTile][kSegs];         // pixels by which the last match of (row, seg) overhangs it
                };
            };
            uint32_t s1[GCN10_N_RASTERS], s2[GCN10_N_RASTERS];
            uint32_t sig[GCN10_N_RASTERS], cand[GCN10_N_RASTERS], differ;
            uint32_t alias[GCN10_N_RASTERS];
        } a;
    };
};

__device__ __forceinline__ int run64(unsigned long long m, int p)
{
    // consecutive set bits of m starting at bit p (0 <= p < 64), counted to the word's end
    const unsigned long long inv = ~(m >> p);       // the bits shifted in from above are zeros: inv != 0 unless p == 0 and m is all ones
    return inv ? __builtin_ctzll(inv) : 64;
}

__global__ __launch_bounds__(kFA2Threads, 8) void fused_stats_seg_kernel(const FusedJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedFA2 &sh = *reinterpret_cast<SharedFA2 *>(smem);
    typedef uint32_t u32_u __attribute__((aligned(1)));
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int seg = __builtin_amdgcn_readfirstlane(t >> 8);     // a wave is 64 consecutive rows of one segment
    const int row = t & 255;
    const uint32_t tiles = job.t.across * job.t.down;
    const uint32_t tix = blockIdx.x;
    const uint32_t ty = tix / job.t.across, tx = tix - ty * job.t.across;
    const uint8_t *class_val = job.class_of + gcn10::kClassCodes * 256;
    // (pass B adds the stream sizes up per raster and 64 positions for pass F-C's placement: from zero)
    if (job.t.chunk_tot && tix == 0u)
        for (uint32_t i = (uint32_t)t; i < job.n_sel * job.t.n_chunks; i += blockDim.x)
            job.t.chunk_tot[i] = 0u;
    // (timing experiments, option "fused_stats_stop" = p + 1: leave after phase p; the workspace keeps the results
    // of the last complete launch, so the passes behind this one still see valid input)
    const uint32_t stop_after = (job.diag >> 8) ? (job.diag >> 8) - 1u : 99u;

    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kFA2Threads)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    if (t < kTile) {
        const uint32_t y = ty * kTile + (uint32_t)t;
        uint32_t srow = y < job.t.rows ? (uint32_t)job.cj[y] : 0u;
        sh.soil_row[t] = srow < job.hx_rows ? srow : job.hx_rows - 1u;
    }
    __syncthreads();
    // class tile: thread = 4 columns x 16 rows (rows (t >> 6) + 16 i)
    {
        const uint32_t x = tx * kTile + (uint32_t)lane * 4u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(sh.tile) + lane;
        const int r0 = t >> 6;
        if ((tx + 1) * kTile <= job.t.W && (ty + 1) * kTile <= job.t.rows) {
            const uint8_t *pe = job.esa + ((size_t)ty * kTile + (uint32_t)r0) * job.t.W + x;
            const uint8_t *ph = job.hx + x;
            uint32_t e4[16], c4[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                e4[i] = *reinterpret_cast<const u32_u *>(pe + (size_t)i * 16 * job.t.W);
                c4[i] = *reinterpret_cast<const u32_u *>(ph + (size_t)sh.soil_row[r0 + 16 * i] * job.hx_stride);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t out = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t lc = (e4[i] >> (8 * q)) & 0xffu;
                    const uint32_t cc = compact_code((c4[i] >> (8 * q)) & 0xffu);
                    out |= (uint32_t)sh.class_of[cc * 256u + lc] << (8 * q);
                }
                dst[(r0 + 16 * i) * (kRowStride / 4)] = out;
            }
        }
        else {
#pragma unroll 4
            for (int i = 0; i < 16; i++) {
                const int r = r0 + 16 * i;
                dst[r * (kRowStride / 4)] = class_pixels4(job, x, ty * kTile + (uint32_t)r, sh.class_of);
            }
        }
    }
    __syncthreads();
    if (stop_after == 0u)
        return;                 // phase 0: class map + landcover / soil loads + class tile
    for (int i = t; i < 288; i += kFA2Threads)
        sh.a.lit_hist[i] = 0;
    if (t < 2)
        sh.a.dist_hist[t] = 0;
    if (t < 256) {
        sh.a.n_c[0][t] = sh.a.n_c[1][t] = 0;
        sh.a.w_c[0][t] = sh.a.w_c[1][t] = 0;
    }
    if (t < GCN10_N_RASTERS) {
        sh.a.s1[t] = 0;
        sh.a.s2[t] = 0;
    }

    // the two candidate masks of this lane's 64 pixels
    unsigned long long near_ = 0, far_ = 0;
    const uint8_t *rowp = sh.tile + row * kRowStride;
    {
        const uint32_t *r32 = reinterpret_cast<const uint32_t *>(rowp) + seg * 16;
        const uint32_t *above = r32 - kRowStride / 4;
        uint32_t prev = seg > 0 ? r32[-1] : (row > 0 ? above[63] : 0u);      // (seg 0: the stream's previous byte is the row above's last)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t r = r32[j];
            const uint32_t shifted = (r << 8) | (prev >> 24);
            near_ |= (unsigned long long)zero_bytes(r ^ shifted) << (4 * j);
            if (row > 0)
                far_ |= (unsigned long long)zero_bytes(r ^ above[j]) << (4 * j);
            prev = r;
        }
        if (row == 0 && seg == 0)
            near_ &= ~1ull;             // the tile's first byte has no predecessor
    }
    if (stop_after == 1u)
        return;                 // phase 1: + the two candidate masks
    sh.a.lead_n[row][seg] = (uint8_t)(~near_ ? __builtin_ctzll(~near_) : kSegPx);
    sh.a.lead_f[row][seg] = (uint8_t)(~far_ ? __builtin_ctzll(~far_) : kSegPx);
    __syncthreads();

    // pixels and Adler weights per class, run by run (a run that crosses into the next segment is
    // counted in two pieces: both sums are additive)
    {
        unsigned long long brk = ~near_ | 1ull;
        const uint32_t i0 = (uint32_t)row * kTile + (uint32_t)seg * kSegPx;    // index of the segment's first byte in the tile
        while (brk) {
            const int x0 = __builtin_ctzll(brk);
            brk &= brk - 1ull;
            const int x1 = brk ? __builtin_ctzll(brk) : kSegPx;
            const uint32_t n = (uint32_t)(x1 - x0);
            const uint32_t c = rowp[seg * kSegPx + x0];
            // sum of (65536 - i) over i = i0 + x0 .. i0 + x1 - 1
            atomicAdd(&sh.a.n_c[lane & 1][c], n);
            atomicAdd(&sh.a.w_c[lane & 1][c], n * ((uint32_t)kTileBytes - i0) - ((n * (uint32_t)(x0 + x1 - 1)) >> 1));
        }
    }

    if (stop_after == 2u)
        return;                 // phase 2: + pixels and weights per class, run by run
    // length of the run of `far ? far_ : near_` that starts at pixel p of this segment, followed into the
    // segments behind it
    auto ext_run = [&](bool far, int p) -> int {
        int r = run64(far ? far_ : near_, p);
        if (p + r == kSegPx)
            for (int k = seg + 1; k < kSegs; k++) {
                const int l = far ? sh.a.lead_f[row][k] : sh.a.lead_n[row][k];
                r += l;
                if (l < kSegPx)
                    break;
            }
        return r;
    };
    // where a match can start: a run of >= 3 at either distance (the last two pixels look into the next segment)
    unsigned long long cand;
    {
        const uint32_t ln = seg + 1 < kSegs ? sh.a.lead_n[row][seg + 1] : 0u, lf = seg + 1 < kSegs ? sh.a.lead_f[row][seg + 1] : 0u;
        const unsigned long long n2 = ln >= 2u ? 3ull : ln, f2 = lf >= 2u ? 3ull : lf;
        cand = (near_ & ((near_ >> 1) | (n2 & 1ull) << 63) & ((near_ >> 2) | n2 << 62)) |
               (far_ & ((far_ >> 1) | (f2 & 1ull) << 63) & ((far_ >> 2) | f2 << 62));
    }
    auto span = [](int from, int to) -> unsigned long long {       // bits from .. min(to, 64) - 1
        const unsigned long long hi = to >= kSegPx ? ~0ull : (1ull << to) - 1ull;
        return hi & (~0ull << from);
    };
    // (1) speculative parse from pixel 0
    unsigned long long starts = 0, fars = 0, covered = 0;
    uint32_t exit_over = 0;
    {
        int pos = 0;
        while (pos < kSegPx) {
            const unsigned long long rest = cand >> pos;
            if (!rest)
                break;
            const int p = pos + __builtin_ctzll(rest);
            const int l1 = ext_run(false, p), l256 = ext_run(true, p);
            const bool far = l256 > l1;                 // tie: distance 1 (no extra bits)
            const int len = far ? l256 : l1;            // >= 3: bit p of cand is set
            starts |= 1ull << p;
            fars |= far ? 1ull << p : 0ull;
            covered |= span(p, p + len);
            pos = p + len;
        }
        exit_over = pos > kSegPx ? (uint32_t)(pos - kSegPx) : 0u;
    }
    // (2) the real entry offset, segment by segment
    if (seg == 0)
        sh.a.exit_at[row][0] = (uint16_t)exit_over;
    for (int sgm = 1; sgm < kSegs; sgm++) {
        __syncthreads();
        if (seg != sgm)
            continue;
        const uint32_t over = sh.a.exit_at[row][sgm - 1];
        if (over >= (uint32_t)kSegPx) {
            // the whole segment lies inside a match that started before it
            starts = fars = 0;
            covered = ~0ull;
            exit_over = over - (uint32_t)kSegPx;
        }
        else if (over > 0) {
            const unsigned long long s_tok = ~covered | starts;    // token starts of the speculative parse
            unsigned long long r_starts = 0, r_fars = 0, r_cov = (1ull << over) - 1ull;
            int pos = (int)over;
            for (;;) {
                if (pos >= kSegPx) {                    // no meeting point: the re-parse is the parse
                    starts = r_starts;
                    fars = r_fars;
                    covered = r_cov;
                    exit_over = (uint32_t)(pos - kSegPx);
                    break;
                }
                const unsigned long long tq = s_tok >> pos, rest = cand >> pos;
                const int q = tq ? pos + __builtin_ctzll(tq) : kSegPx, p = rest ? pos + __builtin_ctzll(rest) : kSegPx;
                if (q <= p) {
                    // pixels pos .. q-1 are literals; at q both parses stand on a token start: the same from there on
                    const unsigned long long hi = q < kSegPx ? ~0ull << q : 0ull;
                    starts = r_starts | (starts & hi);
                    fars = r_fars | (fars & hi);
                    covered = r_cov | (covered & hi);
                    if (q >= kSegPx)
                        exit_over = 0;
                    break;
                }
                const int l1 = ext_run(false, p), l256 = ext_run(true, p);
                const bool far = l256 > l1;
                const int len = far ? l256 : l1;
                r_starts |= 1ull << p;
                r_fars |= far ? 1ull << p : 0ull;
                r_cov |= span(p, p + len);
                pos = p + len;
            }
        }
        if (sgm + 1 < kSegs)
            sh.a.exit_at[row][sgm] = (uint16_t)exit_over;
    }
    if (stop_after == 3u)
        return;                 // phase 3: + the parse (speculative + entry offsets)
    // statistics of the final parse
    const unsigned long long lits = ~covered;
    const uint32_t my_tokens = (uint32_t)__popcll(lits) + (uint32_t)__popcll(starts);
    // (the statistics of the tokens -- literals by class, match length symbols, the two distances -- are counted where
    // the tokens are written out, below: one walk over a segment's literals and one over its matches instead of two
    // each; the scan in between needs the counts of tokens only)

    // raster order = (row, seg): an exclusive prefix over the 1024 counts says where each segment's tokens go
    sh.a.seg_at[row * kSegs + seg] = my_tokens;
    __syncthreads();
    {
        const uint32_t mine = sh.a.seg_at[t];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += up;
        }
        if (lane == 63)
            sh.a.wave_tot[t >> 6] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < (t >> 6); w++)
            before += sh.a.wave_tot[w];
        sh.a.seg_at[t] = before + incl - mine;
        __syncthreads();
    }
    {
        uint32_t at = sh.a.seg_at[row * kSegs + seg];
        uint16_t *out = job.tok + (size_t)tix * kTokStride;
        if (!(job.diag & 16u)) {
            // Literals and matches in loops of their own (a token's place is its rank among the segment's token
            // starts): in one loop over all tokens a wave executed both paths in every trip, as often as its busiest
            // lane has tokens -- 18 of the kernel's 76 us on noisy tiles.  (Neither the LDS round trip per token nor
            // the scattered 2-byte stores were what that cost: reading ahead, or writing the tokens side by side,
            // changed nothing.)
            const unsigned long long all = lits | starts;
            for (unsigned long long m = lits; m; m &= m - 1ull) {
                const int p = __builtin_ctzll(m);
                const uint32_t c = rowp[seg * kSegPx + p];
                if (!(job.diag & 8u))
                    atomicAdd(&sh.a.lit_hist[c], 1u);
                out[at + (uint32_t)__popcll(all & ((1ull << p) - 1ull))] = (uint16_t)c;
            }
            for (unsigned long long m = starts; m; m &= m - 1ull) {
                const int p = __builtin_ctzll(m);
                const bool far = ((fars >> p) & 1ull) != 0;
                const int len = ext_run(far, p);
                uint32_t ev;                    // (no table: a global load per match inside this divergent loop)
                const int lc = length_code_extra(len, ev);
                atomicAdd(&sh.a.lit_hist[257 + lc], 1u);
                atomicAdd(&sh.a.dist_hist[far ? 1 : 0], 1u);
                out[at + (uint32_t)__popcll(all & ((1ull << p) - 1ull))] =
                    (uint16_t)(kTokMatch | (uint32_t)lc | ev << 5 | (far ? 1u : 0u) << 10);
            }
            at += my_tokens;
        }
        else {
            // (timing: without the token write-out; the statistics all the same)
            for (unsigned long long m = starts; m; m &= m - 1ull) {
                const int p = __builtin_ctzll(m);
                const bool far = ((fars >> p) & 1ull) != 0;
                atomicAdd(&sh.a.lit_hist[257 + length_code(ext_run(far, p))], 1u);
                atomicAdd(&sh.a.dist_hist[far ? 1 : 0], 1u);
            }
            if (!(job.diag & 8u))
                for (unsigned long long m = lits; m; m &= m - 1ull)
                    atomicAdd(&sh.a.lit_hist[rowp[seg * kSegPx + __builtin_ctzll(m)]], 1u);
            at += my_tokens;
        }
        if (t == kFA2Threads - 1) {
            out[at] = (uint16_t)kTokEnd;
            job.n_tok[tix] = at + 1u;
        }
    }
    __syncthreads();

    if (stop_after == 4u)
        return;                 // phase 4: + statistics, scan, token write-out
    // From here on the tile is no longer needed (its memory holds the value histograms below), and the work is
    // per (raster, class).  The row form does it with 256 threads, raster after raster (18 x 3 wave sums in a
    // row, three rounds of six value histograms); here the 16 waves split the rasters.
    // (a) Adler-32 sums and the (class, value) hash of every raster: wave wv takes rasters wv, wv + 16
    if (t == 0)
        sh.a.differ = 0;
    {
        const int wv = t >> 6;
        for (uint32_t j = (uint32_t)wv; j < job.n_sel; j += kFA2Threads / 64) {
            const uint8_t *val = class_val + job.sel[j] * 256;
            uint32_t a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t c = (uint32_t)lane + 64u * (uint32_t)q;
                const uint32_t n = sh.a.n_c[0][c] + sh.a.n_c[1][c];
                if (n) {
                    const uint32_t v = val[c];
                    uint32_t h = (c * 0x9E3779B1u + v * 0x85EBCA6Bu + 0x27D4EB2Fu) * 0x165667B1u;
                    h ^= h >> 15;
                    a1 += n * v;
                    a2 += ((sh.a.w_c[0][c] % 65521u + sh.a.w_c[1][c] % 65521u) % 65521u) * v;       // (<= 256 x 65520 x 255 < 2^32 over the tile)
                    a3 += h;
                }
            }
            a1 = wave_sum64(a1);
            a2 = wave_sum64(a2);
            a3 = wave_sum64(a3);
            if (lane == 0) {
                sh.a.s1[j] = a1;
                sh.a.s2[j] = a2;
                sh.a.sig[j] = a3;
            }
        }
    }
    const bool cls = t < 256;
    // (b) rasters that cannot differ on this tile (see the row form): the hash proposes, a comparison class by
    // class confirms
    {
        const bool present = cls && (sh.a.n_c[0][t] | sh.a.n_c[1][t]) != 0;
        __syncthreads();
        if ((uint32_t)t < job.n_sel) {
            uint32_t cand = 0xffu;
            for (uint32_t q = 0; q < (uint32_t)t; q++)
                if (cand == 0xffu && sh.a.sig[q] == sh.a.sig[t])
                    cand = q;
            sh.a.cand[t] = cand;
        }
        __syncthreads();
        {
            uint32_t mine = 0;
            if (present)
                for (uint32_t j = 1; j < job.n_sel; j++) {
                    const uint32_t q = sh.a.cand[j];
                    if (q != 0xffu && class_val[job.sel[j] * 256 + t] != class_val[job.sel[q] * 256 + t])
                        mine |= 1u << j;
                }
            if (mine)
                atomicOr(&sh.a.differ, mine);
        }
        __syncthreads();
        if ((uint32_t)t < job.n_sel) {
            const uint32_t q = sh.a.cand[t];
            sh.a.alias[t] = (q != 0xffu && !((sh.a.differ >> t) & 1u)) ? (kAliasFlag | q) : 0u;
        }
    }
    // (c) per raster: literal counts by VALUE, all rasters at once in the memory of the tile
    uint32_t *HV = reinterpret_cast<uint32_t *>(sh.tile);           // [n_sel][256]
    static_assert(GCN10_N_RASTERS * 256 * 4 <= kTile * kRowStride, "value histograms fit the tile's memory");
    for (uint32_t i = (uint32_t)t; i < job.n_sel * 256u; i += kFA2Threads)
        HV[i] = 0;
    __syncthreads();
    {
        const uint32_t c = (uint32_t)t & 255u;
        const uint32_t lits_c = sh.a.lit_hist[c];
        if (lits_c)
            for (uint32_t j = (uint32_t)t >> 8; j < job.n_sel; j += kFA2Threads / 256)
                atomicAdd(&HV[j * 256u + class_val[job.sel[j] * 256 + c]], lits_c);
    }
    __syncthreads();
    for (uint32_t idx = (uint32_t)t; idx < job.n_sel * (uint32_t)kHistWords; idx += kFA2Threads) {
        const uint32_t j = idx / (uint32_t)kHistWords;
        const int i = (int)(idx - j * (uint32_t)kHistWords);
        uint32_t v;
        if (i < 256)
            v = HV[j * 256u + (uint32_t)i];
        else if (i == 256)
            v = 1u;                             // end of block
        else if (i < 288)
            v = sh.a.lit_hist[i];               // match length symbols: the same for every raster
        else if (i < 290)
            v = sh.a.dist_hist[i - 288];
        else if (i == 290)
            v = (((65536u % 65521u + sh.a.s2[j] % 65521u) % 65521u) << 16) | ((1u + sh.a.s1[j]) % 65521u);
        else if (i == 291)
            v = sh.a.alias[j];
        else
            v = 0u;
        job.t.hist[((size_t)j * tiles + tix) * kHistWords + i] = v;
    }
}

constexpr int kStageWords = 88;         // 64 tokens x 41 bits, starting anywhere in the first word

struct SharedFC {
    union {
        struct {
            uint32_t cl[288][8];            // code | length << 16 of the group's rasters, by class / length symbol
            uint32_t stage[4][kGroup][kStageWords];     // per wave: the bits of its 64 tokens, per raster
        } c;
        uint8_t class_of[gcn10::kClassCodes * 256];     // stored fallback only, after the streams
    };
    uint32_t wave_tot[kGroup / 2][4];       // bits of each wave's tokens, two streams per word
    uint32_t masks[2];
};

// pass F-C: one workgroup per (tile position, group of kGroup rasters), token-parallel.
// Every thread takes one token of the position's stream per trip, looks up the group's codes
// for it (one LDS read), a prefix sum over the workgroup gives its bit position in each of
// the kGroup streams, the bits are OR-ed into a per-wave staging area in LDS and go to the
// arena as whole words (the first and last word of a wave's span are OR-ed into the zeroed
// slot, the words in between are plain coalesced stores).  No row walk, no divergence: the
// work is the number of tokens, whatever their distribution over the rows.
__global__ __launch_bounds__(kTile) void fused_emit_kernel(const FusedJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedFC &sh = *reinterpret_cast<SharedFC *>(smem);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: selects on it stay scalar
    const uint32_t tiles = job.t.across * job.t.down;
    const uint32_t tix = blockIdx.x;
    const uint32_t ty = tix / job.t.across, tx = tix - ty * job.t.across;
    const uint32_t j0 = blockIdx.y * kGroup;
    const uint32_t nj = job.n_sel - j0 < (uint32_t)kGroup ? job.n_sel - j0 : (uint32_t)kGroup;
    const uint8_t *class_val = job.class_of + gcn10::kClassCodes * 256;

    // which rasters of the group have a slot, and which of those are stored
    uint32_t live, stored_mask;
    {
        int mine = 0, st = 0;
        if ((uint32_t)t < nj) {
            const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)(j0 + t) * tiles + tix) * kBookBytes);
            mine = job.t.table[((size_t)(j0 + t) * tiles + tix) * 2] < kAliasSlot;      // neither "arena too small" nor an alias
            st = mine && b->stream_bytes == (uint32_t)kMaxStream;
            // an alias's table entry is its original's (written by pass B, a launch ago)
            const uint32_t al = job.t.hist[((size_t)(j0 + t) * tiles + tix) * kHistWords + 291];
            if (al & kAliasFlag) {
                const size_t from = ((size_t)(al & 0xffu) * tiles + tix) * 2, to = ((size_t)(j0 + t) * tiles + tix) * 2;
                job.t.table[to] = job.t.table[from];
                job.t.table[to + 1] = job.t.table[from + 1];
            }
        }
        live = (uint32_t)__ballot(mine);            // threads t < nj are all in wave 0: its ballots are the
        stored_mask = (uint32_t)__ballot(st);       // masks, the other waves get them through LDS
        if (t == 0) {
            sh.masks[0] = live;
            sh.masks[1] = stored_mask;
        }
        __syncthreads();
        live = sh.masks[0];
        stored_mask = sh.masks[1];
        if (live == 0)
            return;
    }
    const uint32_t coded = live & ~stored_mask;     // rasters that get a Huffman stream

    // codes of the group's rasters by class (a literal of class c is the symbol val(c)), the
    // zeroed slots with their block headers, and per raster what is the same for every token
    uint32_t *words[kGroup];
    uint32_t base[kGroup], dcode0[kGroup], dlen0[kGroup], dcode1[kGroup], dlen1[kGroup];
#pragma unroll
    for (int k = 0; k < kGroup; k++) {
        words[k] = nullptr;
        base[k] = dcode0[k] = dlen0[k] = dcode1[k] = dlen1[k] = 0;
        if (!((coded >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        words[k] = reinterpret_cast<uint32_t *>(job.t.arena + job.t.table[((size_t)j * tiles + tix) * 2]);
        base[k] = b->header_bits;
        dcode0[k] = b->dist_code[0];
        dlen0[k] = b->dist_len[0];
        dcode1[k] = (uint32_t)b->dist_code[1] | 63u << b->dist_len[1];      // + 6 extra bits: 256 - 193
        dlen1[k] = (uint32_t)b->dist_len[1] + 6u;
        // (up to the slot's 16-byte end: the host writes a raster's streams of a strip as one extent,
        // the few bytes between two streams included -- they are zeros, not whatever the arena held)
        const uint32_t n_words = (b->stream_bytes + 15u) / 16u * 4u;
        for (uint32_t i = t; i < n_words; i += kTile)
            words[k][i] = i < 64u ? b->header[i] : 0u;
        for (int i = t; i < 288; i += kTile) {
            const uint32_t sym = i < 256 ? class_val[job.sel[j] * 256 + i] : (uint32_t)i;
            sh.c.cl[i][k] = (uint32_t)b->lit_code[sym] | (uint32_t)b->lit_len[sym] << 16;
        }
    }
    for (int i = t; i < 4 * kGroup * kStageWords; i += kTile)
        (&sh.c.stage[0][0][0])[i] = 0u;
    __threadfence_block();                          // the zeroed slots are in place before this workgroup ORs into them
    __syncthreads();

    if (coded && !(job.diag & 2u)) {
        const uint16_t *tok = job.tok + (size_t)tix * kTokStride;
        const uint32_t n_tok = job.n_tok[tix];
        for (uint32_t i0 = 0; i0 < n_tok; i0 += kTile) {
            const uint32_t i = i0 + (uint32_t)t;
            const bool have = i < n_tok;
            const uint32_t tk = have ? tok[i] : 0u;
            const bool is_match = (tk & kTokMatch) != 0;
            const uint32_t sym = is_match ? 257u + (tk & 31u) : (tk & kTokEnd) ? 256u : (tk & 255u);
            const uint32_t lc = tk & 31u;
            const uint32_t ne = !is_match || lc < 8u || lc == 28u ? 0u : (lc - 4u) >> 2;
            const uint32_t ev = is_match ? (tk >> 5) & 31u : 0u;
            const bool far = ((tk >> 10) & 1u) != 0;
            const u32x4 c03 = *reinterpret_cast<const u32x4 *>(&sh.c.cl[sym][0]);
            const u32x4 c47 = *reinterpret_cast<const u32x4 *>(&sh.c.cl[sym][4]);
            // the token's bits in every stream of the group: at most 15 + 5 + 15 + 6 = 41
            uint32_t lo[kGroup], hi[kGroup], n[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t c = k < 4 ? c03[k] : c47[k - 4];
                const uint32_t l = c >> 16;
                const uint32_t nb = l + ne;                         // <= 20
                const uint32_t d = is_match ? (far ? dcode1[k] : dcode0[k]) : 0u;
                const uint32_t dl = is_match ? (far ? dlen1[k] : dlen0[k]) : 0u;
                lo[k] = (c & 0xffffu) | ev << l | d << nb;
                hi[k] = (d >> 1) >> (31u - nb);
                n[k] = have && ((coded >> k) & 1u) ? nb + dl : 0u;
            }
            // prefix sums over the wave, two streams per register (a wave's total is < 2^16)
            uint32_t excl[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; k += 2) {
                const uint32_t mine = n[k] | n[k + 1] << 16;
                const uint32_t incl = wave_scan_dpp(mine);
                const uint32_t ex = incl - mine;
                excl[k] = ex & 0xffffu;
                excl[k + 1] = ex >> 16;
                if (lane == 63)
                    sh.wave_tot[k / 2][wave] = incl;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                if (!((coded >> k) & 1u))
                    continue;
                const int h = 16 * (k & 1);
                const uint32_t t0 = (sh.wave_tot[k / 2][0] >> h) & 0xffffu, t1 = (sh.wave_tot[k / 2][1] >> h) & 0xffffu,
                               t2 = (sh.wave_tot[k / 2][2] >> h) & 0xffffu, t3 = (sh.wave_tot[k / 2][3] >> h) & 0xffffu;
                const uint32_t before = wave == 0 ? 0u : wave == 1 ? t0 : wave == 2 ? t0 + t1 : t0 + t1 + t2;
                const uint32_t mine = wave == 0 ? t0 : wave == 1 ? t1 : wave == 2 ? t2 : t3;
                const uint32_t first = base[k] + before;            // this wave's first bit in the stream
                const uint32_t word0 = first >> 5;
                uint32_t *stage = sh.c.stage[wave][k];
                {
                    // 41 bits shifted by up to 31: three words, OR-ed unconditionally (absent tokens are zeros)
                    const uint32_t rel = (first & 31u) + excl[k];
                    const uint32_t w = rel >> 5, sft = rel & 31u;
                    const uint32_t l32 = n[k] ? lo[k] : 0u, h32 = n[k] ? hi[k] : 0u;
                    atomicOr(&stage[w], l32 << sft);
                    atomicOr(&stage[w + 1], (l32 >> 1) >> (31u - sft) | h32 << sft);
                    atomicOr(&stage[w + 2], (h32 >> 1) >> (31u - sft));
                }
                // (one wave: the LDS atomics above are complete before the reads below issue)
                const uint32_t n_words = ((first & 31u) + mine + 31u) >> 5;
                for (uint32_t w = (uint32_t)lane; w < n_words; w += 64u) {
                    const uint32_t val = stage[w];
                    stage[w] = 0u;
                    if (w == 0u || w == n_words - 1u)
                        atomicOr(&words[k][word0 + w], val);
                    else
                        words[k][word0 + w] = val;
                }
                base[k] += t0 + t1 + t2 + t3;
            }
            __syncthreads();                        // wave_tot is rewritten in the next trip
        }
    }
    // trailers: the Adler-32 of the raster's tile, big endian, after the last (padded) byte
    if ((uint32_t)t < nj && ((coded >> t) & 1u)) {
        const uint32_t j = j0 + (uint32_t)t;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        const uint32_t adler = job.t.hist[((size_t)j * tiles + tix) * kHistWords + 290];
        uint32_t *w = reinterpret_cast<uint32_t *>(job.t.arena + job.t.table[((size_t)j * tiles + tix) * 2]);
        const uint32_t at = b->stream_bytes - 4u;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t byte = (adler >> (24 - 8 * i)) & 0xffu;
            atomicOr(&w[(at + i) >> 2], byte << (8 * ((at + i) & 3u)));
        }
    }
    if (stored_mask == 0)
        return;

    // stored fallback (incompressible tiles): the raster's bytes are val(class), two blocks of
    // 32768 bytes; the classes are formed again from landcover + soil
    __syncthreads();
    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kTile)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    __syncthreads();
    for (uint32_t k = 0; k < nj; k++) {
        if (!((stored_mask >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        uint8_t *o = job.t.arena + job.t.table[((size_t)j * tiles + tix) * 2];
        const uint32_t adler = job.t.hist[((size_t)j * tiles + tix) * kHistWords + 290];
        if (t == 0) {
            o[0] = 0x78;
            o[1] = 0x01;
            for (int blk = 0; blk < 2; blk++) {
                uint8_t *h = o + 2 + blk * (5 + 32768);
                h[0] = (uint8_t)(blk == 1);
                h[1] = 0x00;
                h[2] = 0x80;
                h[3] = 0xff;
                h[4] = 0x7f;
            }
            const uint32_t at = (uint32_t)kMaxStream - 4u;
            o[at] = (uint8_t)(adler >> 24);
            o[at + 1] = (uint8_t)(adler >> 16);
            o[at + 2] = (uint8_t)(adler >> 8);
            o[at + 3] = (uint8_t)adler;
        }
        const uint8_t *val = class_val + job.sel[j] * 256;
        const uint32_t xc = (uint32_t)(t & 63) * 4u;
        for (int i = 0; i < kTile / 4; i++) {
            const int r = i * 4 + (t >> 6);
            const uint32_t c4 = class_pixels4(job, tx * kTile + xc, ty * kTile + (uint32_t)r, sh.class_of);
            uint8_t *dst = o + 2 + (r >> 7) * (5 + 32768) + 5 + (r & 127) * kTile + xc;
            dst[0] = val[c4 & 0xffu];
            dst[1] = val[(c4 >> 8) & 0xffu];
            dst[2] = val[(c4 >> 16) & 0xffu];
            dst[3] = val[c4 >> 24];
        }
    }
}

// ------------------------------------------------------------------------
// pass F-C, wave-independent form (round 3, the default).  Same streams as fused_emit_kernel, bit for
// bit.  That kernel walks the token stream 256 tokens per trip with all four waves in lock step, one
// token per lane, and ORs every token's bits into LDS staging words with three 32-bit atomics per
// stream.  Timing switches (profiles/r03) put its time in two places: LDS atomics whose lanes hit the
// same word (a token is ~8 bits: eight neighbouring lanes per 64-bit word, executed one after the
// other) and the write-out of ~20 staged words per stream and trip behind them -- with the next trip's
// token load queued behind those stores (vmcnt is one in-order queue).  Here
//   * every wave owns one contiguous QUARTER of the position's tokens: phase 1 adds up the bits its
//     quarter takes in each of the group's streams (no packing), ONE workgroup barrier, and the sums
//     before it say where the wave's bits begin in every stream; nothing is shared after that;
//   * a lane takes FOUR consecutive tokens and packs their codes into a register chunk (<= 164 bits)
//     before anything touches LDS: a wave's trip is 256 tokens, a lane's chunk is ~40 bits, so at most
//     two lanes meet in a 64-bit staging word and there are a quarter of the atomics;
//   * the group's six codes of a class are ONE 16-byte LDS row (20 bits per stream: code | length << 15);
//   * the next trip's tokens are loaded before this trip's words are stored.
// ------------------------------------------------------------------------
constexpr int kLaneToks = 4;                        // consecutive tokens per lane and trip
constexpr int kTripToks = 64 * kLaneToks;           // 256 tokens per wave and trip
constexpr int kStage64 = (63 + kTripToks * 41 + 63) / 64 + 2;      // a trip's bits starting anywhere in the first word: 166 words

struct SharedFC2 {
    union {
        struct {
            unsigned long long cl[288][2];                          // 20 bits per stream: code | length << 15; streams 0-2, 3-5
            unsigned long long stage[4][kStage64];                  // per wave: the bits of one trip of one stream
        } c;
        uint8_t class_of[gcn10::kClassCodes * 256];                 // stored fallback only, after the streams
    };
    uint32_t chunk_bits[kGroup][4];         // bits of each wave's quarter, per stream
    uint32_t masks[2];
    uint32_t r_tot[GCN10_N_RASTERS];        // arena bytes of every raster's streams of this strip (16-byte slots)
    uint32_t r_pre[GCN10_N_RASTERS];        // ... of its positions before this one
    uint32_t slot[kGroup];                  // where this position's streams of the group lie
};

// what a 16-bit token says: symbol, extra bits of a length, which distance
struct TokFields {
    uint32_t sym, ne, ev;
    bool is_match, far;
};

__device__ __forceinline__ TokFields tok_fields(uint32_t tk)
{
    TokFields f;
    f.is_match = (tk & kTokMatch) != 0;
    const uint32_t lc = tk & 31u;
    f.sym = f.is_match ? 257u + lc : (tk & kTokEnd) ? 256u : (tk & 255u);
    f.ne = !f.is_match || lc < 8u || lc == 28u ? 0u : (lc - 4u) >> 2;
    f.ev = f.is_match ? (tk >> 5) & 31u : 0u;
    f.far = ((tk >> 10) & 1u) != 0;
    return f;
}

// code | length << 15 of stream k (0..5) from the two halves of a cl row
__device__ __forceinline__ uint32_t cl_field(unsigned long long h0, unsigned long long h1, int k)
{
    return (uint32_t)((k < 3 ? h0 : h1) >> (20 * (k % 3))) & 0xfffffu;
}

__global__ __launch_bounds__(kTile, 6) void fused_emit_wave_kernel(const FusedJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedFC2 &sh = *reinterpret_cast<SharedFC2 *>(smem);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t tiles = job.t.across * job.t.down;
    const uint32_t tix = blockIdx.x;
    const uint32_t ty = tix / job.t.across, tx = tix - ty * job.t.across;
    const uint32_t j0 = blockIdx.y * kGroup;
    const uint32_t nj = job.n_sel - j0 < (uint32_t)kGroup ? job.n_sel - j0 : (uint32_t)kGroup;
    const uint8_t *class_val = job.class_of + gcn10::kClassCodes * 256;

    // Where this position's streams go (round 3: pass B' folded in -- a launch of one workgroup and its kernel
    // boundary were 16-18 us per strip).  Pass B left sizes[i] = { alias mark or 0, stream bytes } and added every
    // stream's 16-byte-aligned size to chunk_tot[raster][position / 64].  The layout is raster-major, tiles of a
    // raster in order, 16-byte slots, every raster's extent starting at a multiple of seg_align (a power of two):
    // the offset of (raster j, this position) is the sum of the aligned totals of the rasters before j plus the sizes
    // of raster j's earlier positions.  Every workgroup adds those up for itself: the chunk totals of the rasters up
    // to its own (one load per lane for a 1024-row strip of a 3-degree block: 18 x 9 values) and the sizes of its own
    // chunk of positions (at most 4.5 loads per lane) -- one memory latency, against the 61 KB table a workgroup
    // would otherwise have to read.  It writes the offsets of ITS six streams into the table for the host and keeps
    // them in LDS for itself.  The workgroups of position 0 zero the pads in front of their rasters' extents; the one
    // of the last group also writes the bytes used.
    {
        const uint32_t jn = j0 + nj;                    // (an alias's original is an earlier raster, maybe of another group)
        const uint32_t chunks = job.t.n_chunks, c = tix >> 6;
        if (t < GCN10_N_RASTERS) {
            sh.r_tot[t] = 0;
            sh.r_pre[t] = 0;
        }
        __syncthreads();
        for (uint32_t i = (uint32_t)t; i < jn * chunks; i += (uint32_t)kTile) {
            const uint32_t v = job.t.chunk_tot[i];
            const uint32_t r = i / chunks, cc = i - r * chunks;
            if (v) {
                atomicAdd(&sh.r_tot[r], v);
                if (cc < c)
                    atomicAdd(&sh.r_pre[r], v);
            }
        }
        const uint2 *tab = reinterpret_cast<const uint2 *>(job.t.sizes);
        for (uint32_t r = (uint32_t)wave; r < jn; r += 4u) {
            const uint32_t pp = c * 64u + (uint32_t)lane;
            uint32_t nd = 0;
            if (pp < tix) {
                const uint2 e = tab[(size_t)r * tiles + pp];
                nd = e.x == kAliasSlot ? 0u : (e.y + (uint32_t)(kSlotAlign - 1)) & ~(uint32_t)(kSlotAlign - 1);
            }
            nd = wave_scan_dpp(nd);
            if (lane == 63 && nd)
                atomicAdd(&sh.r_pre[r], nd);
        }
    }
    __syncthreads();
    // which rasters of the group have a slot, and which of those are stored
    uint32_t live, stored_mask;
    {
        int mine = 0, st = 0;
        if (wave == 0) {
            // extents: seg[j] = sum over r < j of the aligned totals (a scan over lanes 0..17)
            const unsigned long long am = (unsigned long long)job.seg_align - 1ull;
            const unsigned long long tot = (uint32_t)lane < job.n_sel ? sh.r_tot[lane] : 0u;
            unsigned long long incl = (tot + am) & ~am;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const unsigned long long up = __shfl_up(incl, off, 64);
                if (lane >= off)
                    incl += up;
            }
            const unsigned long long seg = incl - ((tot + am) & ~am);      // where raster `lane`'s extent starts
            const bool own = (uint32_t)t < nj;
            const uint32_t j = own ? j0 + (uint32_t)t : 0u;
            const uint32_t al = own ? job.t.hist[((size_t)j * tiles + tix) * kHistWords + 291] : 0u;
            // an alias's table entry is its original's; the original may belong to another group's workgroup, so its
            // offset is worked out here as well
            const uint32_t src_j = al & kAliasFlag ? (al & 0xffu) : j;
            const unsigned long long src_seg = __shfl(seg, (int)src_j, 64);
            if (own) {
                const size_t to = ((size_t)j * tiles + tix) * 2;
                const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
                const uint32_t bytes = al & kAliasFlag ? job.t.sizes[((size_t)src_j * tiles + tix) * 2 + 1] : b->stream_bytes;
                const unsigned long long off = src_seg + sh.r_pre[src_j];
                const uint32_t need = (bytes + (uint32_t)(kSlotAlign - 1)) & ~(uint32_t)(kSlotAlign - 1);
                const bool fits = off + need <= job.t.arena_cap;
                job.t.table[to] = fits ? (uint32_t)off : 0xffffffffu;
                job.t.table[to + 1] = fits ? bytes : 0u;
                sh.slot[t] = fits ? (uint32_t)off : 0xffffffffu;
                mine = !(al & kAliasFlag) && fits;       // neither "arena too small" nor an alias
                st = mine && bytes == (uint32_t)kMaxStream;
            }
            if (tix == 0u) {
                // pads: behind each of my rasters' last stream up to the next extent's start (behind the last raster:
                // up to the next multiple of the alignment: an O_DIRECT write reads that far); *cursor = bytes used
                for (uint32_t k = 0; k < nj; k++) {
                    const uint32_t r = j0 + k;
                    const unsigned long long from = __shfl(seg, (int)r, 64) + __shfl(tot, (int)r, 64);
                    unsigned long long to = (from + am) & ~am;
                    if (to > job.t.arena_cap)
                        to = job.t.arena_cap;
                    for (unsigned long long o = from + (unsigned long long)lane * 16u; o + 16u <= to; o += 64u * 16u)
                        *reinterpret_cast<u32x4 *>(job.t.arena + o) = u32x4{ 0u, 0u, 0u, 0u };
                    if (r + 1u == job.n_sel && lane == 0)
                        *job.t.cursor = from;
                }
            }
        }
        live = (uint32_t)__ballot(mine);
        stored_mask = (uint32_t)__ballot(st);
        if (t == 0) {
            sh.masks[0] = live;
            sh.masks[1] = stored_mask;
        }
        __syncthreads();
        live = sh.masks[0];
        stored_mask = sh.masks[1];
        if (live == 0)
            return;
    }
    const uint32_t coded = __builtin_amdgcn_readfirstlane(live & ~stored_mask);     // rasters that get a Huffman stream

    // per stream: where it lies, where its tokens begin, its two distance codes (wave-uniform: scalar registers)
    unsigned long long *words[kGroup];
    uint32_t base[kGroup], dcode0[kGroup], dlen0[kGroup], dcode1[kGroup], dlen1[kGroup];
    for (int i = t; i < 288 * 2; i += kTile)
        (&sh.c.cl[0][0])[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGroup; k++) {
        words[k] = nullptr;
        base[k] = dcode0[k] = dlen0[k] = dcode1[k] = dlen1[k] = 0;
        if (!((coded >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        uint32_t *w32 = reinterpret_cast<uint32_t *>(job.t.arena + sh.slot[k]);
        words[k] = reinterpret_cast<unsigned long long *>(w32);
        base[k] = __builtin_amdgcn_readfirstlane(b->header_bits);
        dcode0[k] = __builtin_amdgcn_readfirstlane((uint32_t)b->dist_code[0]);
        dlen0[k] = __builtin_amdgcn_readfirstlane((uint32_t)b->dist_len[0]);
        dcode1[k] = __builtin_amdgcn_readfirstlane((uint32_t)b->dist_code[1] | 63u << b->dist_len[1]);     // + 6 extra bits: 256 - 193
        dlen1[k] = __builtin_amdgcn_readfirstlane((uint32_t)b->dist_len[1] + 6u);
        // the zeroed slot with its block header, up to the slot's 16-byte end (the host writes a raster's
        // streams of a strip as one extent: the bytes between two streams are zeros)
        const uint32_t n_words = (b->stream_bytes + 15u) / 16u * 4u;
        for (uint32_t i = t; i < n_words; i += kTile)
            w32[i] = i < 64u ? b->header[i] : 0u;
        // (k-th 20-bit field of its half: every (i, k) is touched by one thread, the halves were zeroed above)
        for (int i = t; i < 288; i += kTile) {
            const uint32_t sym = i < 256 ? class_val[job.sel[j] * 256 + i] : (uint32_t)i;
            const unsigned long long f = (unsigned long long)((uint32_t)b->lit_code[sym] | (uint32_t)b->lit_len[sym] << 15);
            sh.c.cl[i][k / 3] |= f << (20 * (k % 3));
        }
    }
    for (int i = t; i < 4 * kStage64; i += kTile)
        (&sh.c.stage[0][0])[i] = 0ull;

    const uint16_t *tok = job.tok + (size_t)tix * kTokStride;
    const uint32_t n_tok = coded && !(job.diag & 2u) ? job.n_tok[tix] : 0u;
    // this wave's quarter: whole trips of 256 tokens
    const uint32_t per_wave = ((n_tok + 4u * kTripToks - 1u) / (4u * kTripToks)) * kTripToks;
    const uint32_t c0 = (uint32_t)wave * per_wave;
    const uint32_t c1 = c0 + per_wave < n_tok ? c0 + per_wave : n_tok;
    __threadfence_block();                          // the zeroed slots are in place before this workgroup ORs into them
    __syncthreads();

    // four consecutive tokens of this lane (8-byte aligned: kTokStride and the trips are multiples of 4)
    auto load4 = [&](uint32_t i0) -> uint2 {
        const uint32_t i = i0 + (uint32_t)lane * kLaneToks;
        return i < c1 ? *reinterpret_cast<const uint2 *>(tok + i) : make_uint2(0u, 0u);
    };
    auto tok_of = [](uint2 v, int q) -> uint32_t { return ((q < 2 ? v.x : v.y) >> (16 * (q & 1))) & 0xffffu; };

    // phase 1: the bits of this quarter in every stream
    {
        uint32_t acc[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; k++)
            acc[k] = 0;
        for (uint32_t i0 = (job.diag & 1u) ? c1 : c0; i0 < c1; i0 += kTripToks) {      // (diag 1: timing without phase 1)
            const uint2 v = load4(i0);
#pragma unroll
            for (int q = 0; q < kLaneToks; q++) {
                if (i0 + (uint32_t)lane * kLaneToks + (uint32_t)q >= c1)
                    continue;
                const TokFields f = tok_fields(tok_of(v, q));
                const unsigned long long h0 = sh.c.cl[f.sym][0], h1 = sh.c.cl[f.sym][1];
#pragma unroll
                for (int k = 0; k < kGroup; k++)
                    acc[k] += (cl_field(h0, h1, k) >> 15) + f.ne + (f.is_match ? (f.far ? dlen1[k] : dlen0[k]) : 0u);
            }
        }
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            const uint32_t tot = wave_scan_dpp(acc[k]);
            if (lane == 63)
                sh.chunk_bits[k][wave] = (coded >> k) & 1u ? tot : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGroup; k++) {
        const uint32_t b0 = sh.chunk_bits[k][0], b1 = sh.chunk_bits[k][1], b2 = sh.chunk_bits[k][2];
        base[k] += wave == 0 ? 0u : wave == 1 ? b0 : wave == 2 ? b0 + b1 : b0 + b1 + b2;
        base[k] = __builtin_amdgcn_readfirstlane(base[k]);
    }

    // phase 2: 256 tokens per trip, this wave alone
    unsigned long long *stage = sh.c.stage[wave];
    unsigned long long carry[kGroup];               // per stream: the unfinished last word of the previous trip (wave-uniform)
#pragma unroll
    for (int k = 0; k < kGroup; k++)
        carry[k] = 0ull;
    uint2 next = load4(c0);
    for (uint32_t i0 = c0; i0 < c1; i0 += kTripToks) {
        const uint2 v = next;
        next = load4(i0 + kTripToks);               // (in front of this trip's stores: vmcnt is one in-order queue)
        TokFields f[kLaneToks];
        unsigned long long h0[kLaneToks], h1[kLaneToks];
        bool have[kLaneToks];
#pragma unroll
        for (int q = 0; q < kLaneToks; q++) {
            have[q] = i0 + (uint32_t)lane * kLaneToks + (uint32_t)q < c1;
            f[q] = tok_fields(have[q] ? tok_of(v, q) : 0u);         // (past the end: a harmless symbol, no bits)
            h0[q] = sh.c.cl[f[q].sym][0];
            h1[q] = sh.c.cl[f[q].sym][1];
        }
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            if (!((coded >> k) & 1u))
                continue;
            // the lane's four tokens packed into a chunk of <= 4 x 41 = 164 bits
            unsigned long long a0 = 0, a1 = 0, a2 = 0;
            uint32_t off = 0;
#pragma unroll
            for (int q = 0; q < kLaneToks; q++) {
                const uint32_t c = cl_field(h0[q], h1[q], k);
                const uint32_t l = c >> 15;
                const uint32_t nb = l + f[q].ne;                        // <= 20
                const uint32_t d = f[q].is_match ? (f[q].far ? dcode1[k] : dcode0[k]) : 0u;
                const uint32_t dl = f[q].is_match ? (f[q].far ? dlen1[k] : dlen0[k]) : 0u;
                const uint32_t n = have[q] ? nb + dl : 0u;
                const unsigned long long bits = n ? (unsigned long long)((c & 0x7fffu) | f[q].ev << l) | (unsigned long long)d << nb : 0ull;
                // off <= 41 q: the token starts in word 0 or 1 of the chunk
                const uint32_t s = off & 63u;
                const unsigned long long lo = bits << s, hi = (bits >> 1) >> (63u - s);
                if (q == 0) {
                    a0 = bits;
                }
                else if (q == 1) {                  // off <= 41
                    a0 |= lo;
                    a1 |= hi;
                }
                else {                              // off <= 123
                    const bool w1 = off >= 64u;
                    a0 |= w1 ? 0ull : lo;
                    a1 |= w1 ? lo : hi;
                    a2 |= w1 ? hi : 0ull;
                }
                off += n;
            }
            // where the chunk goes: prefix sum of the lanes' bit counts (a trip's total is < 2^16)
            const uint32_t incl = wave_scan_dpp(off);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t first = base[k];                             // this trip's first bit in the stream (wave-uniform)
            // the unfinished last word of this stream's previous trip is the first word of this one
            if (lane == 0)
                stage[0] = carry[k];
            if (!(job.diag & 4u)) {                                     // (diag 4: timing without the LDS atomics)
                const uint32_t rel = (first & 63u) + incl - off;
                const uint32_t w = rel >> 6, s = rel & 63u;
                const unsigned long long v0 = a0 << s, v1 = a1 << s | (a0 >> 1) >> (63u - s),
                                         v2 = a2 << s | (a1 >> 1) >> (63u - s), v3 = (a2 >> 1) >> (63u - s);
                if (v0) atomicOr(&stage[w], v0);
                if (v1) atomicOr(&stage[w + 1], v1);
                if (v2) atomicOr(&stage[w + 2], v2);
                if (v3) atomicOr(&stage[w + 3], v3);
            }
            // (one wave: LDS executes its instructions in order, the reads below see the atomics above)
            // Words of this trip: the last one, if the trip does not end on a word boundary, stays with the wave
            // (carry) -- except in the wave's last trip -- so that only the first word of a wave's first trip and
            // the last word of its last trip are shared with anyone (the header, a neighbouring wave, the trailer)
            // and need a global atomic; everything else is a plain 8-byte store.  (Two atomics per word-sharing
            // trip were 0.10 of this kernel's 0.15 ms per noisy strip: profiles/r03.)
            const uint32_t end = (first & 63u) + total;
            const uint32_t n_words = (end + 63u) >> 6;                  // <= 165
            const bool last_trip = i0 + kTripToks >= c1;
            const bool keep_last = (end & 63u) != 0u && !last_trip;     // the last word is unfinished and this wave goes on
            const uint32_t n_out = keep_last ? n_words - 1u : n_words;
            unsigned long long kept = 0;
            if (!(job.diag & 16u))                                      // (diag 16: timing without the write-out)
                for (uint32_t w = (uint32_t)lane; w < n_words; w += 64u) {
                    const unsigned long long val = stage[w];
                    stage[w] = 0ull;
                    if (w >= n_out) {
                        kept = val;                 // one lane
                        continue;
                    }
                    unsigned long long *dst = words[k] + (first >> 6) + w;
                    if (job.diag & 32u)             // (diag 32: timing with the LDS half of the write-out only)
                        continue;
                    const bool shared = (w == 0u && i0 == c0) || (w == n_words - 1u && last_trip);
                    if (shared && !(job.diag & 8u))     // (diag 8: timing with plain stores only)
                        atomicOr(dst, val);
                    else
                        *dst = val;
                }
            if (keep_last) {
                const uint32_t src = (n_words - 1u) & 63u;
                carry[k] = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kept, (int)src) |
                           (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kept >> 32), (int)src) << 32;
            }
            else {
                carry[k] = 0ull;
            }
            base[k] = first + total;
        }
    }
    // trailers: the Adler-32 of the raster's tile, big endian, after the last (padded) byte
    if ((uint32_t)t < nj && ((coded >> t) & 1u)) {
        const uint32_t j = j0 + (uint32_t)t;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        const uint32_t adler = job.t.hist[((size_t)j * tiles + tix) * kHistWords + 290];
        uint32_t *w = reinterpret_cast<uint32_t *>(job.t.arena + sh.slot[t]);
        const uint32_t at = b->stream_bytes - 4u;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t byte = (adler >> (24 - 8 * i)) & 0xffu;
            atomicOr(&w[(at + i) >> 2], byte << (8 * ((at + i) & 3u)));
        }
    }
    if (stored_mask == 0)
        return;

    // stored fallback (incompressible tiles): the raster's bytes are val(class), two blocks of
    // 32768 bytes; the classes are formed again from landcover + soil
    __syncthreads();
    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kTile)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    __syncthreads();
    for (uint32_t k = 0; k < nj; k++) {
        if (!((stored_mask >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        uint8_t *o = job.t.arena + sh.slot[k];
        const uint32_t adler = job.t.hist[((size_t)j * tiles + tix) * kHistWords + 290];
        if (t == 0) {
            o[0] = 0x78;
            o[1] = 0x01;
            for (int blk = 0; blk < 2; blk++) {
                uint8_t *h = o + 2 + blk * (5 + 32768);
                h[0] = (uint8_t)(blk == 1);
                h[1] = 0x00;
                h[2] = 0x80;
                h[3] = 0xff;
                h[4] = 0x7f;
            }
            const uint32_t at = (uint32_t)kMaxStream - 4u;
            o[at] = (uint8_t)(adler >> 24);
            o[at + 1] = (uint8_t)(adler >> 16);
            o[at + 2] = (uint8_t)(adler >> 8);
            o[at + 3] = (uint8_t)adler;
        }
        const uint8_t *val = class_val + job.sel[j] * 256;
        const uint32_t xc = (uint32_t)(t & 63) * 4u;
        for (int i = 0; i < kTile / 4; i++) {
            const int r = i * 4 + (t >> 6);
            const uint32_t c4 = class_pixels4(job, tx * kTile + xc, ty * kTile + (uint32_t)r, sh.class_of);
            uint8_t *dst = o + 2 + (r >> 7) * (5 + 32768) + 5 + (r & 127) * kTile + xc;
            dst[0] = val[c4 & 0xffu];
            dst[1] = val[(c4 >> 8) & 0xffu];
            dst[2] = val[(c4 >> 16) & 0xffu];
            dst[3] = val[c4 >> 24];
        }
    }
}

}  // namespace

extern "C" {

int gcn10_gpu_deflate_fused_available(gcn10_gpu_ctx *ctx)
{
    return ctx && ctx->n_tables > 0 && ctx->n_classes > 0 ? 1 : 0;
}

int gcn10_gpu_deflate_fused_strip(gcn10_gpu_ctx *ctx, const uint8_t *esa, int W, int rows, const int32_t *cj,
                                  unsigned cond_mask, unsigned table_mask, uint8_t *arena_dev, size_t arena_cap,
                                  uint32_t *table_dev, unsigned long long *cursor_dev, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (ctx->n_tables == 0)
        return fail(GCN10_E_STATE, "gcn10_gpu_deflate_fused_strip: call gcn10_gpu_set_tables first");
    if (ctx->n_classes == 0)
        return fail(GCN10_E_STATE, "gcn10_gpu_deflate_fused_strip: the lookup tables define more than 256 pixel "
                                   "classes; use gcn10_gpu_cn_strip + gcn10_gpu_deflate_strip");
    if (!ctx->d_hx || ctx->hx_W == 0 || (uint32_t)W != ctx->hx_W)
        return fail(GCN10_E_STATE, "gcn10_gpu_deflate_fused_strip: call gcn10_gpu_prepare_tile for W=%d first", W);
    if (W <= 0 || rows < 0 || cond_mask == 0 || (cond_mask & ~3u) || table_mask == 0 ||
        (table_mask >> ctx->n_tables))
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_fused_strip: bad shape or masks");
    if (rows == 0)
        return GCN10_OK;
    if (!esa || !cj || !arena_dev || !table_dev || !cursor_dev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_fused_strip: null pointer");
    if ((reinterpret_cast<uintptr_t>(arena_dev) & 15u) != 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_fused_strip: arena must be 16-byte aligned");

    FusedJob job;
    memset(&job, 0, sizeof job);
    job.esa = esa;
    job.hx = ctx->d_hx;
    job.cj = cj;
    job.class_of = ctx->d_class_of;
    job.hx_stride = ctx->hx_stride;
    job.hx_rows = ctx->hx_rows;
    job.diag = (uint32_t)ctx->fused_diag | (uint32_t)ctx->fused_stats_stop << 8;
    job.seg_align = (uint32_t)ctx->arena_segment_align;
    for (int r = 0; r < GCN10_N_RASTERS; r++)
        if ((cond_mask >> (r / 9)) & 1u && (table_mask >> (r % 9)) & 1u)
            job.sel[job.n_sel++] = (uint8_t)r;
    job.t.arena = arena_dev;
    job.t.table = table_dev;
    job.t.cursor = cursor_dev;
    job.t.W = (uint32_t)W;
    job.t.rows = (uint32_t)rows;
    job.t.across = ((uint32_t)W + kTile - 1) / kTile;
    job.t.down = ((uint32_t)rows + kTile - 1) / kTile;
    // stream offsets are 32-bit table entries with 0xfffffffe / 0xffffffff reserved
    if (arena_cap >= 0xfffffffeull)
        return fail(GCN10_E_INVAL, "tile encoder: an arena of %zu bytes does not fit 32-bit stream offsets "
                                   "(use fewer rows per strip)", arena_cap);
    job.t.arena_cap = arena_cap;
    job.t.codes_stop = (uint32_t)ctx->codes_stop;
    const uint32_t positions = job.t.across * job.t.down;
    // (pass F-C adds a raster's stream sizes up in 32 bits; with the 4 GB arena limit above only a strip far beyond
    // any arena can get here)
    if ((uint64_t)positions * (uint64_t)(kMaxStream + kSlotAlign) >= 0xffffffffull)
        return fail(GCN10_E_INVAL, "tile encoder: %u tile positions in one strip (use fewer rows per strip)", positions);
    const uint64_t nblocks = (uint64_t)positions * job.n_sel;
    job.t.n_tiles = (uint32_t)nblocks;

    // workspace: statistics + code books per (raster, tile), token tiles + match starts per position
    const size_t stats_bytes = ((size_t)nblocks * ((size_t)kHistWords * 4 + (size_t)kBookBytes) + 255) & ~(size_t)255;
    const size_t tok_bytes = (size_t)positions * kTokStride * sizeof(uint16_t);
    const size_t ntok_bytes = ((size_t)positions * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t need = stats_bytes + tok_bytes + ntok_bytes + (size_t)nblocks * 8 + (size_t)job.n_sel * ((positions + 63u) / 64u) * 4;
    rc = gcn10::deflate_workspace(ctx, need);
    if (rc)
        return rc;
    job.t.hist = reinterpret_cast<uint32_t *>(ctx->deflate_ws);
    job.t.books = reinterpret_cast<uint8_t *>(ctx->deflate_ws) + (size_t)nblocks * kHistWords * 4;
    job.tok = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(ctx->deflate_ws) + stats_bytes);
    job.n_tok = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->deflate_ws) + stats_bytes + tok_bytes);
    // the wave-independent pass F-C places the streams itself, every workgroup from all the sizes: those must not be
    // the table the workgroups write the offsets to
    job.t.sizes = table_dev;
    if (ctx->fused_emit != 0) {
        job.t.sizes = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->deflate_ws) + stats_bytes + tok_bytes + ntok_bytes);
        job.t.chunk_tot = job.t.sizes + (size_t)nblocks * 2;        // zeroed by pass F-A, added to by pass B
        job.t.n_chunks = (positions + 63u) / 64u;
    }

    static_assert(sizeof(SharedFA) <= 80 * 1024, "two fused statistics workgroups per CU");
    static_assert(sizeof(SharedFA2) <= 80 * 1024, "two segment-parallel statistics workgroups (32 waves) per CU");
    static_assert(sizeof(SharedFC) <= 20 * 1024, "eight fused emit workgroups per CU");
    static_assert(sizeof(SharedFC2) <= 16 * 1024, "ten wave-independent emit workgroups per CU by LDS");
    if (!ctx->fused_ready) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stats_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFA)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stats_seg_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFA2)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_emit_wave_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFC2)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_emit_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFC)));
        ctx->fused_ready = true;
    }
    hipStream_t s = as_stream(ctx, stream);
    if (ctx->fused_parse == 0)
        hipLaunchKernelGGL(fused_stats_kernel, dim3(positions), dim3(kTile), sizeof(SharedFA), s, job);
    else
        hipLaunchKernelGGL(fused_stats_seg_kernel, dim3(positions), dim3(kFA2Threads), sizeof(SharedFA2), s, job);
    // (pass B' as a launch of its own only for the lock-step form of pass F-C: the wave-independent form places its streams itself)
    rc = gcn10::deflate_launch_codes(ctx, job.t, (uint32_t)nblocks, s, ctx->fused_emit == 0);
    if (rc)
        return rc;
    const uint32_t groups = (job.n_sel + kGroup - 1) / kGroup;
    if (ctx->fused_emit == 0)
        hipLaunchKernelGGL(fused_emit_kernel, dim3(positions, groups), dim3(kTile), sizeof(SharedFC), s, job);
    else
        hipLaunchKernelGGL(fused_emit_wave_kernel, dim3(positions, groups), dim3(kTile), sizeof(SharedFC2), s, job);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

}  // extern "C"

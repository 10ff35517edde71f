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
// little compression.  Passes: F-A statistics per tile position -> B code construction
// per (raster, tile), unchanged -> F-C emission per tile position, looping over rasters.
// ------------------------------------------------------------------------
constexpr int kGroup = 6;            // rasters emitted by one workgroup of pass F-C

struct FusedJob {
    const uint8_t *esa;
    const uint8_t *hx;
    const int32_t *cj;
    const uint8_t *class_of;        // [36][256]; class_val [18][256] follows
    uint8_t *tok;                   // [positions][kTileBytes] tokenised class tiles (F-A -> F-C)
    unsigned long long *tok_start;  // [positions][kTile][4]   where their match tokens start
    uint32_t hx_stride, hx_rows;
    uint32_t diag;                  // gcn10_gpu_set_option("fused_diag"): timing experiments
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

// Class ids of one tile position into LDS (row stride kRowStride).
__device__ __forceinline__ void load_class_tile(const FusedJob &job, uint32_t tx, uint32_t ty,
                                                const uint8_t *class_of_lds, uint8_t *tile, int t)
{
    const uint32_t x = tx * kTile + (uint32_t)(t & 63) * 4u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(tile) + (t & 63);
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
__device__ __forceinline__ void tokenise_row(uint8_t *tile, int t, const RowMasks &m, uint32_t *lit_hist,
                                             uint32_t *dist_hist, unsigned long long (&start)[4])
{
    uint8_t *row = tile + t * kRowStride;
    int x = 0;
    start[0] = start[1] = start[2] = start[3] = 0ull;
    while (x < kTile) {
        const int cand = next_candidate(m, x);
        for (; x < cand; x++)
            atomicAdd(&lit_hist[row[x]], 1u);
        if (x >= kTile)
            break;
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
}

__device__ __forceinline__ uint32_t wave_sum64(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

struct SharedFA {
    uint8_t tile[kTile * kRowStride];
    union {
        uint8_t class_of[gcn10::kClassCodes * 256];     // while the tile is built
        struct {
            uint32_t lit_hist[288];     // literals by class, match length symbols at 257..
            uint32_t dist_hist[2];
            uint32_t n_c[256], w_c[256];                // pixels per class, sum of their Adler weights
            uint32_t H[kGroup][256];                    // literal counts by VALUE, kGroup rasters at a time
            uint32_t s1[GCN10_N_RASTERS], s2[GCN10_N_RASTERS];
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

    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kTile)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    __syncthreads();
    load_class_tile(job, tx, ty, sh.class_of, sh.tile, t);
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

    // pixels and Adler weights per class, run by run along row t (weight of byte i: 65536 - i)
    {
        const uint8_t *row = sh.tile + t * kRowStride;
        uint32_t cur = row[0], n = 1;
        uint32_t wgt = (uint32_t)kTileBytes - (uint32_t)t * kTile;
        uint32_t wsum = wgt;
        for (int x = 1; x < kTile; x++) {
            const uint32_t c = row[x];
            wgt--;
            if (c == cur) {
                n++;
                wsum += wgt;
            }
            else {
                atomicAdd(&sh.a.n_c[cur], n);
                atomicAdd(&sh.a.w_c[cur], wsum);
                cur = c;
                n = 1;
                wsum = wgt;
            }
        }
        atomicAdd(&sh.a.n_c[cur], n);
        atomicAdd(&sh.a.w_c[cur], wsum);
    }
    unsigned long long start[4];
    tokenise_row(sh.tile, t, m, sh.a.lit_hist, sh.a.dist_hist, start);
    {
        unsigned long long *dst = job.tok_start + ((size_t)tix * kTile + t) * 4;
        dst[0] = start[0];
        dst[1] = start[1];
        dst[2] = start[2];
        dst[3] = start[3];
    }
    __syncthreads();
    // the token tile -> workspace, a wave per row and step
    {
        uint32_t *dst = reinterpret_cast<uint32_t *>(job.tok + (size_t)tix * kTileBytes) + (t & 63);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(sh.tile) + (t & 63);
#pragma unroll 8
        for (int i = 0; i < kTile / 4; i++) {
            const int r = i * 4 + (t >> 6);
            dst[r * (kTile / 4)] = src[r * (kRowStride / 4)];
        }
    }

    // Adler-32 of every raster's tile: thread t = class t, s1 = 1 + sum n_c val, s2 = N + sum w_c val
    const uint32_t lits = sh.a.lit_hist[t];
    {
        const uint32_t n = sh.a.n_c[t];
        const uint32_t w = sh.a.w_c[t] % 65521u;
        for (uint32_t j = 0; j < job.n_sel; j++) {
            const uint32_t v = class_val[job.sel[j] * 256 + t];
            const uint32_t p1 = wave_sum64(n * v);              // <= 2^24 in all
            const uint32_t p2 = wave_sum64(w * v);              // <= 256 * 65520 * 255 < 2^32 in all
            if ((t & 63) == 0) {
                atomicAdd(&sh.a.s1[j], p1);
                atomicAdd(&sh.a.s2[j], p2);
            }
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
                else
                    v = 0u;
                out[i] = v;
            }
        }
        __syncthreads();
    }
}

struct SharedFC {
    uint8_t tile[kTile * kRowStride];   // token tile
    union {
        struct {
            uint32_t cl[288][8];            // code | length << 16 of the group's rasters, by class / length symbol
            uint32_t lenpack[288][kGroup / 2];  // the lengths alone, two rasters per dword (row measure)
        } c;
        uint8_t class_of[gcn10::kClassCodes * 256];     // stored fallback only, after the walk
    };
    uint32_t wave_sum[kGroup][4];
};

// A row's bits go straight into the stream's words in the arena: the words a row fills
// completely are plain stores, its first and last (shared with the neighbouring rows, the
// header or the trailer) are OR-ed into the zeroed slot.
struct WordEmitter {
    uint32_t *words;
    uint32_t wpos;
    unsigned long long acc;
    uint32_t nacc;
    bool first;
    bool dry;
    __device__ __forceinline__ void init(uint32_t *w, uint32_t start_bit)
    {
        dry = false;
        words = w;
        wpos = start_bit >> 5;
        nacc = start_bit & 31u;
        acc = 0ull;
        first = true;
    }
    __device__ __forceinline__ void put(uint32_t value, uint32_t nbits)
    {
        acc |= (unsigned long long)value << nacc;
        nacc += nbits;
        if (nacc >= 32u) {
            if (dry)
                ;
            else if (first)
                atomicOr(&words[wpos], (uint32_t)acc);
            else
                words[wpos] = (uint32_t)acc;
            first = false;
            wpos++;
            acc >>= 32;
            nacc -= 32u;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc > 0u && !dry)
            atomicOr(&words[wpos], (uint32_t)acc & (0xffffffffu >> (32u - nacc)));
    }
};

// pass F-C: one workgroup per (tile position, group of kGroup rasters).  The token tile is
// loaded once; one walk over each row measures it for all rasters of the group (packed
// 16-bit sums), a prefix sum places the rows, and ONE more walk emits the row for all
// rasters of the group at once: per token one LDS read of the group's codes, kGroup bit
// accumulators in registers, words written straight to the arena (no stream image in LDS).
__global__ __launch_bounds__(kTile) void fused_emit_kernel(const FusedJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedFC &sh = *reinterpret_cast<SharedFC *>(smem);
    const int t = threadIdx.x;
    const uint32_t tiles = job.t.across * job.t.down;
    const uint32_t tix = blockIdx.x;
    const uint32_t ty = tix / job.t.across, tx = tix - ty * job.t.across;
    const uint32_t j0 = blockIdx.y * kGroup;
    const uint32_t nj = job.n_sel - j0 < (uint32_t)kGroup ? job.n_sel - j0 : (uint32_t)kGroup;
    const uint8_t *class_val = job.class_of + gcn10::kClassCodes * 256;

    // which rasters of the group have a slot, and which of those are stored
    uint32_t live = 0, stored_mask = 0;
    {
        int mine = 0, st = 0;
        if ((uint32_t)t < nj) {
            const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)(j0 + t) * tiles + tix) * kBookBytes);
            mine = b->slot != 0xffffffffu;
            st = mine && b->stream_bytes == (uint32_t)kMaxStream;
        }
        live = (uint32_t)__ballot(mine);            // threads t < nj are all in wave 0: its ballots are the
        stored_mask = (uint32_t)__ballot(st);       // masks, the other waves get them through LDS
        if (t == 0) {
            sh.wave_sum[0][0] = live;
            sh.wave_sum[0][1] = stored_mask;
        }
        __syncthreads();
        live = sh.wave_sum[0][0];
        stored_mask = sh.wave_sum[0][1];
        __syncthreads();
        if (live == 0)
            return;
    }
    const uint32_t coded = live & ~stored_mask;     // rasters that get a Huffman stream

    // token tile and this row's match starts
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(job.tok + (size_t)tix * kTileBytes) + (t & 63);
        uint32_t *dst = reinterpret_cast<uint32_t *>(sh.tile) + (t & 63);
        uint32_t v[kTile / 4];
#pragma unroll
        for (int i = 0; i < kTile / 4; i++)
            v[i] = src[(i * 4 + (t >> 6)) * (kTile / 4)];
#pragma unroll
        for (int i = 0; i < kTile / 4; i++)
            dst[(i * 4 + (t >> 6)) * (kRowStride / 4)] = v[i];
    }
    unsigned long long start[4];
    {
        const unsigned long long *src = job.tok_start + ((size_t)tix * kTile + t) * 4;
        start[0] = src[0];
        start[1] = src[1];
        start[2] = src[2];
        start[3] = src[3];
    }
    // codes of the group's rasters by class (a literal of class c is the symbol val(c)), the
    // zeroed slots with their block headers, and per raster what is the same for every row
    uint32_t *words[kGroup];
    uint32_t header_bits[kGroup], dcode0[kGroup], dlen0[kGroup], dcode1[kGroup], dlen1[kGroup];
    uint32_t dist_pack[2][kGroup / 2] = {};
#pragma unroll
    for (int k = 0; k < kGroup; k++) {
        words[k] = nullptr;
        header_bits[k] = dcode0[k] = dlen0[k] = dcode1[k] = dlen1[k] = 0;
        if (!((coded >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        words[k] = reinterpret_cast<uint32_t *>(job.t.arena + b->slot);
        header_bits[k] = b->header_bits;
        dcode0[k] = b->dist_code[0];
        dlen0[k] = b->dist_len[0];
        dcode1[k] = (uint32_t)b->dist_code[1] | 63u << b->dist_len[1];      // + 6 extra bits: 256 - 193
        dlen1[k] = (uint32_t)b->dist_len[1] + 6u;
        dist_pack[0][k / 2] |= dlen0[k] << (16 * (k & 1));
        dist_pack[1][k / 2] |= dlen1[k] << (16 * (k & 1));
        const uint32_t n_words = (b->stream_bytes + 3u) / 4u;
        for (uint32_t i = t; i < n_words; i += kTile)
            words[k][i] = i < 64u ? b->header[i] : 0u;
        for (int i = t; i < 288; i += kTile) {
            const uint32_t sym = i < 256 ? class_val[job.sel[j] * 256 + i] : (uint32_t)i;
            sh.c.cl[i][k] = (uint32_t)b->lit_code[sym] | (uint32_t)b->lit_len[sym] << 16;
        }
    }
    __syncthreads();
    for (int i = t; i < 288; i += kTile) {
#pragma unroll
        for (int k = 0; k < kGroup; k += 2) {
            const uint32_t lo = ((coded >> k) & 1u) ? sh.c.cl[i][k] >> 16 : 0u;
            const uint32_t hi = ((coded >> (k + 1)) & 1u) ? sh.c.cl[i][k + 1] >> 16 : 0u;
            sh.c.lenpack[i][k / 2] = lo | hi << 16;
        }
    }
    __threadfence();                                // the zeroed slots are in place before any row ORs into them
    __syncthreads();

    if (coded) {
        const uint8_t *row = sh.tile + t * kRowStride;
        // bits of row t in every raster of the group
        uint32_t acc[kGroup / 2] = {};
        {
            int x = (job.diag & 4u) ? kTile : 0;
            while (x < kTile) {
                const int p = next_set(start, x);
                for (; x < p; x++) {
                    const uint32_t *lp = sh.c.lenpack[row[x]];
#pragma unroll
                    for (int k = 0; k < kGroup / 2; k++)
                        acc[k] += lp[k];
                }
                if (x >= kTile)
                    break;
                const uint32_t b0 = row[x], b1 = row[x + 1], b2 = row[x + 2];
                const uint32_t *lp = sh.c.lenpack[257u + (b0 & 31u)];
                const uint32_t far = b0 >> 7;
                const uint32_t common = (b1 >> 5) * 0x00010001u;
#pragma unroll
                for (int k = 0; k < kGroup / 2; k++)
                    acc[k] += lp[k] + common + (far ? dist_pack[1][k] : dist_pack[0][k]);
                x += (int)b2 + 3;
            }
        }
        // exclusive prefix over rows, all rasters of the group at once
        uint32_t first_bit[kGroup];
        {
            uint32_t v[kGroup];
            const int lane = t & 63;
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t mine = (acc[k / 2] >> (16 * (k & 1))) & 0xffffu;
                uint32_t s = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = __shfl_up(s, off, 64);
                    if (lane >= off)
                        s += up;
                }
                if (lane == 63)
                    sh.wave_sum[k][t >> 6] = s;
                v[k] = s - mine;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                uint32_t base = 0;
                for (int w = 0; w < (t >> 6); w++)
                    base += sh.wave_sum[k][w];
                first_bit[k] = v[k] + base;
            }
        }
        // the row, for every raster of the group in one walk
        WordEmitter em[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            em[k].init(words[k], header_bits[k] + first_bit[k]);
            em[k].dry = (job.diag & 1u) != 0;
        }
        int x = (job.diag & 2u) ? kTile : 0;
        while (x < kTile) {
            const int p = next_set(start, x);
            for (; x < p; x++) {
                const uint32_t *cl = sh.c.cl[row[x]];
#pragma unroll
                for (int k = 0; k < kGroup; k++) {
                    if ((coded >> k) & 1u) {
                        const uint32_t c = cl[k];
                        em[k].put(c & 0xffffu, c >> 16);
                    }
                }
            }
            if (x >= kTile)
                break;
            const uint32_t b0 = row[x], b1 = row[x + 1], b2 = row[x + 2];
            const uint32_t *cl = sh.c.cl[257u + (b0 & 31u)];
            const bool far = (b0 >> 7) != 0;
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                if ((coded >> k) & 1u) {
                    const uint32_t c = cl[k];
                    const uint32_t l = c >> 16;
                    em[k].put((c & 0xffffu) | (b1 & 31u) << l, l + (b1 >> 5));      // <= 15 + 5 bits
                    em[k].put(far ? dcode1[k] : dcode0[k], far ? dlen1[k] : dlen0[k]);  // <= 15 + 6 bits
                }
            }
            x += (int)b2 + 3;
        }
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            if ((coded >> k) & 1u) {
                if (t == kTile - 1) {
                    const uint32_t c = sh.c.cl[256][k];
                    em[k].put(c & 0xffffu, c >> 16);                               // end of block
                }
                em[k].finish();
            }
        }
    }
    // trailers: the Adler-32 of the raster's tile, big endian, after the last (padded) byte
    if ((uint32_t)t < nj && ((coded >> t) & 1u)) {
        const uint32_t j = j0 + (uint32_t)t;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        const uint32_t adler = job.t.hist[((size_t)j * tiles + tix) * kHistWords + 290];
        uint32_t *w = reinterpret_cast<uint32_t *>(job.t.arena + b->slot);
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
    // 32768 bytes; the classes are formed again from landcover + soil (the tile in LDS holds tokens)
    __syncthreads();
    for (int i = t; i < gcn10::kClassCodes * 256 / 4; i += kTile)
        reinterpret_cast<uint32_t *>(sh.class_of)[i] = reinterpret_cast<const uint32_t *>(job.class_of)[i];
    __syncthreads();
    for (uint32_t k = 0; k < nj; k++) {
        if (!((stored_mask >> k) & 1u))
            continue;
        const uint32_t j = j0 + k;
        const Book *b = reinterpret_cast<const Book *>(job.t.books + ((size_t)j * tiles + tix) * kBookBytes);
        uint8_t *o = job.t.arena + b->slot;
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
    job.diag = (uint32_t)ctx->fused_diag;
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
    job.t.arena_cap = arena_cap;
    const uint32_t positions = job.t.across * job.t.down;
    const uint64_t nblocks = (uint64_t)positions * job.n_sel;
    job.t.n_tiles = (uint32_t)nblocks;

    // workspace: statistics + code books per (raster, tile), token tiles + match starts per position
    const size_t stats_bytes = ((size_t)nblocks * ((size_t)kHistWords * 4 + (size_t)kBookBytes) + 255) & ~(size_t)255;
    const size_t need = stats_bytes + (size_t)positions * ((size_t)kTileBytes + (size_t)kTile * 32);
    rc = gcn10::deflate_workspace(ctx, need);
    if (rc)
        return rc;
    job.t.hist = reinterpret_cast<uint32_t *>(ctx->deflate_ws);
    job.t.books = reinterpret_cast<uint8_t *>(ctx->deflate_ws) + (size_t)nblocks * kHistWords * 4;
    job.tok = reinterpret_cast<uint8_t *>(ctx->deflate_ws) + stats_bytes;
    job.tok_start = reinterpret_cast<unsigned long long *>(job.tok + (size_t)positions * kTileBytes);

    static_assert(sizeof(SharedFA) <= 80 * 1024, "two fused statistics workgroups per CU");
    static_assert(sizeof(SharedFC) <= 80 * 1024, "two fused emit workgroups per CU");
    if (!ctx->fused_ready) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stats_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFA)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_emit_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedFC)));
        ctx->fused_ready = true;
    }
    hipStream_t s = as_stream(ctx, stream);
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(fused_stats_kernel, dim3(positions), dim3(kTile), sizeof(SharedFA), s, job);
    rc = gcn10::deflate_launch_codes(ctx, job.t, (uint32_t)nblocks, s);
    if (rc)
        return rc;
    const uint32_t groups = (job.n_sel + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(fused_emit_kernel, dim3(positions, groups), dim3(kTile), sizeof(SharedFC), s, job);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

}  // extern "C"

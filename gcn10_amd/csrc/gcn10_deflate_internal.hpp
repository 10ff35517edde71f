// gcn10_deflate_internal.hpp -- shared by the two tile encoders of libgcn10_gpu.so:
// gcn10_deflate.hip (one zlib stream per tile of a CN raster held in HBM) and
// gcn10_deflate_fused.hip (the 18 streams of a tile position straight from landcover + soil).
// Constants, the hand-over structures between the passes, the match-candidate bit masks, and
// the launch of pass B (code construction), which both encoders use unchanged.
#ifndef GCN10_DEFLATE_INTERNAL_HPP
#define GCN10_DEFLATE_INTERNAL_HPP

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gcn10_gpu.h"
#include "gcn10_gpu_internal.hpp"

namespace gcn10_deflate {

constexpr int kTile = 256;
constexpr int kRowStride = 260;                 // 65 dwords: rows start in different LDS banks
constexpr int kTileBytes = kTile * kTile;
constexpr int kOutWords = 16416;                // 65 664 B: stored fallback (65 552 B) fits
constexpr int kMaxStream = 2 + 2 * 5 + kTileBytes + 4;     // stored: header, 2 blocks, adler
constexpr int kSlotAlign = 16;
// streams up to this size are emitted by the two-workgroups-per-CU variant of pass C
constexpr int kSmallStream = 12800;
constexpr int kNumLit = 286;
constexpr int kNumDist = 30;

// length 3..258 -> length code 0..28 (symbol 257 + code), RFC 1951 3.2.5
static __device__ const uint8_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                          35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 0 /*258*/ };
static __device__ const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                           3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
// order in which code-length-code lengths are sent, RFC 1951 3.2.7
static __device__ const uint8_t kClOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
// the fixed, complete code for the code-length alphabet: symbols 0-9,16,17,18 get 4 bits,
// 10-15 get 5 bits (13/16 + 6/32 = 1).  Canonical codes, already bit-reversed.
static __device__ const uint8_t kClLen[19] = { 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 4, 4, 4 };

__device__ __forceinline__ uint32_t bitrev(uint32_t code, int len)
{
    return __builtin_bitreverse32(code) >> (32 - len);
}

__device__ __forceinline__ int length_code(int len)
{
    // len in 3..258
    if (len == 258)
        return 28;
    if (len <= 10)
        return len - 3;
    const int l = len - 3;
    const int hb = 31 - __builtin_clz(l);       // floor(log2(l)), >= 3
    const int eb = hb - 2;                      // extra bits
    return 4 * eb + 4 + ((l >> eb) & 3);
}

// length_code() and the value of the code's extra bits (len - kLenBase[code]) without the table: for len > 10,
// len - 3 = (4 + q) << eb | extra with q = two bits, so the extra bits are the low eb bits of len - 3
__device__ __forceinline__ int length_code_extra(int len, uint32_t &extra)
{
    extra = 0;
    if (len == 258)
        return 28;
    if (len <= 10)
        return len - 3;
    const int l = len - 3;
    const int eb = 29 - __builtin_clz(l);
    extra = (uint32_t)l & ((1u << eb) - 1u);
    return 4 * eb + 4 + ((l >> eb) & 3);
}

struct TileJob {
    const uint8_t *const *rasters;      // device array of raster strip pointers
    uint8_t *arena;
    uint32_t *table;                    // [n_rasters][tiles][2] = offset, size
    uint32_t *sizes;                    // [n_rasters][tiles][2] = alias mark or 0, bytes: what pass B leaves for the placement
                                        // (the table itself when pass B' places; a workspace array when pass F-C does)
    uint32_t *chunk_tot;                // [n_rasters][n_chunks] or null: 16-byte-aligned stream bytes of each 64 positions of a raster
    uint32_t n_chunks;                  // (pass F-C places from these: fused encoder, wave-independent emit)
    unsigned long long *cursor;
    uint32_t *hist;                     // [n_tiles][kHistWords]   (pass A -> B)
    uint8_t *books;                     // [n_tiles][kBookBytes]   (pass B -> C)
    uint32_t W, rows, across, down, n_tiles;
    unsigned long long arena_cap;
    uint32_t codes_stop;                // timing experiments only (option "codes_stop"): pass B leaves after phase (value - 1)
};

// per-tile statistics written by pass A: 288 literal/length counts, 2 distance
// counts (codes 0 and 15), the tile's Adler-32
constexpr int kHistWords = 288 + 2 + 2;
// per-tile code book written by pass B
struct Book {
    uint8_t lit_len[288];
    uint16_t lit_code[288];
    uint8_t dist_len[2];        // distance codes 0 (distance 1) and 15 (distance 256)
    uint8_t pad[2];
    uint16_t dist_code[2];
    uint32_t header_bits;       // bit position after the block header (zlib header included)
    uint32_t stream_bytes;      // size of the finished zlib stream (kMaxStream: stored fallback)
    uint32_t slot;              // its offset in the arena, 0xffffffff if the arena is too small
    uint32_t header[64];        // the first header_bits bits of the stream
};
constexpr int kBookBytes = (int)sizeof(Book);
// hist[291] of a (raster, tile): bit 0 = one-value tile (per-raster encoder: value << 8), bit 31 = this
// raster's tile is byte for byte the tile of selected raster (hist[291] & 0xff) -- fused encoder
constexpr uint32_t kAliasFlag = 0x80000000u;
constexpr uint32_t kAliasSlot = 0xfffffffeu;    // Book::slot of an alias (0xffffffff: the arena is too small)

// ------------------------------------------------------------------------
// match candidates of a tile row as bit masks (passes A and C of both encoders)
// ------------------------------------------------------------------------

// 4 flag bits of a dword: bit k set iff byte k of d is zero
__device__ __forceinline__ uint32_t zero_bytes(uint32_t d)
{
    const uint32_t m = ~(((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d | 0x7f7f7f7fu);   // 0x80 per zero byte
    return (((m >> 7) * 0x00204081u) >> 21) & 0xfu;
}

// The two match candidates of every position of tile row t as bit masks:
// near bit x: byte x equals the previous byte of the stream (distance 1)
// far  bit x: byte x equals the byte above it (distance 256)
struct RowMasks {
    unsigned long long near_[4], far_[4];
};

__device__ __forceinline__ void row_masks(const uint8_t *tile, int t, RowMasks &m)
{
    const uint32_t *row = reinterpret_cast<const uint32_t *>(tile + t * kRowStride);
    const uint32_t *above = reinterpret_cast<const uint32_t *>(tile + (t - 1) * kRowStride);
    uint32_t prev = t > 0 ? above[63] : 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned long long nm = 0, fm = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t r = row[q * 16 + j];
            const uint32_t shifted = (r << 8) | (prev >> 24);
            nm |= (unsigned long long)zero_bytes(r ^ shifted) << (4 * j);
            if (t > 0)
                fm |= (unsigned long long)zero_bytes(r ^ above[q * 16 + j]) << (4 * j);
            prev = r;
        }
        m.near_[q] = nm;
        m.far_[q] = fm;
    }
    if (t == 0)
        m.near_[0] &= ~1ull;        // the tile's first byte has no predecessor
}

__device__ __forceinline__ unsigned long long pick(const unsigned long long (&m)[4], int w)
{
    return w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3];
}

// number of consecutive set bits starting at bit x (0 <= x < 256)
__device__ __forceinline__ int run_from(const unsigned long long (&m)[4], int x)
{
    int len = 0;
    for (;;) {
        const int pos = x + len;
        if (pos >= kTile)
            return len;
        const int b = pos & 63;
        const unsigned long long inv = ~(pick(m, pos >> 6) >> b);
        const int n = inv ? __builtin_ctzll(inv) : 64;
        if (n < 64 - b)
            return len + n;
        len += 64 - b;
    }
}

// next position >= x where either mask has a set bit (256 if none)
__device__ __forceinline__ int next_candidate(const RowMasks &m, int x)
{
    for (int pos = x; pos < kTile;) {
        const int b = pos & 63;
        const unsigned long long v = (pick(m.near_, pos >> 6) | pick(m.far_, pos >> 6)) >> b;
        if (v)
            return pos + __builtin_ctzll(v);
        pos += 64 - b;
    }
    return kTile;
}


// Inclusive prefix sum over the 64 lanes with DPP adds (no LDS traffic): three shifted adds of
// the input give sums over 4 lanes, row_shr:4 / row_shr:8 complete the rows of 16, row_bcast:15
// and row_bcast:31 carry the row totals on.
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t x)
{
    uint32_t r = x;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);     // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, false);     // row_shr:4, lanes 4..15
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, false);     // row_shr:8, lanes 8..15
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return r;
}


}  // namespace gcn10_deflate

namespace gcn10 {
using gcn10_deflate::TileJob;

// Grows the context's encoder workspace (statistics, code books, token tiles) to `need` bytes.
int deflate_workspace(gcn10_gpu_ctx *ctx, size_t need);
// Pass B for `nblocks` (raster, tile) pairs whose statistics are in job.hist: code books to job.books,
// sizes to job.table; with `place`, pass B' (deflate_place_kernel) behind it lays the streams out in the arena.
int deflate_launch_codes(gcn10_gpu_ctx *ctx, const TileJob &job, uint32_t nblocks, hipStream_t s, bool place = true);

}  // namespace gcn10

#endif

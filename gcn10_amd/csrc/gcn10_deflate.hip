// gcn10_deflate.hip -- zlib/DEFLATE encoding of CN raster tiles on the GPU.
//
// What it replaces: GDAL's GTiff driver deflates every 256x256 block on the
// host inside save_raster()'s GDALRasterIO (/root/reference/src/raster.c:204-219,
// COMPRESS=DEFLATE, TILED=YES); the reference's paper names that as the cost
// that dominates a run (paper/paper.md:152-153).  With 18 rasters of 1.3 GB per
// block the raw rasters would also have to cross PCIe (23 GB per block).  Here
// the CN strips never leave HBM uncompressed: one workgroup per tile emits a
// complete zlib stream (RFC 1950 wrapper, one RFC 1951 dynamic-Huffman block,
// Adler-32), streams are packed into an arena, and only the arena is copied to
// the host, which appends the tiles to the GeoTIFFs.  Any inflate decodes them;
// the bytes differ from zlib's own output (file bytes are not a parity target,
// decoded pixels are: tests inflate every tile and compare).
//
// Encoder, three launches per strip (all tiles of all 18 rasters in each):
//   matches   only two distances are tried: 1 (run of the previous byte) and 256
//             (same column, row above) -- the two ways CN rasters repeat (10 m
//             landcover patches, 250 m soil cells); a match never crosses the end
//             of its row, so the 256 rows of a tile parse independently (greedy,
//             minimum length 3).  Candidates are found with byte-compare bit masks,
//             so literal stretches and run lengths cost a few bit operations
//   pass A    one workgroup per tile (thread t = row t): symbol statistics, Adler-32
//   pass B    one THREAD per tile: length-limited (15) canonical Huffman code of the
//             tile's own statistics + the dynamic block header; the code-length
//             alphabet uses a fixed complete code (13 x 4 bit, 6 x 5 bit).  Serial
//             per tile; the ~10^4 tiles of a strip supply the parallelism
//   pass C    one workgroup per tile: rows re-parse, sum their bit lengths, a prefix
//             sum places them, rows OR their bits into an LDS image of the stream;
//             stored-block fallback when Huffman coding would exceed the raw size
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "gcn10_gpu.h"
#include "gcn10_gpu_internal.hpp"

using gcn10::as_stream;
using gcn10::fail;
using gcn10::u32x4;
using gcn10::use_device;

namespace {

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
__device__ const uint8_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                          35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 0 /*258*/ };
__device__ const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                           3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
// order in which code-length-code lengths are sent, RFC 1951 3.2.7
__device__ const uint8_t kClOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
// the fixed, complete code for the code-length alphabet: symbols 0-9,16,17,18 get 4 bits,
// 10-15 get 5 bits (13/16 + 6/32 = 1).  Canonical codes, already bit-reversed.
__device__ const uint8_t kClLen[19] = { 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 4, 4, 4 };

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

struct TileJob {
    const uint8_t *const *rasters;      // device array of raster strip pointers
    uint8_t *arena;
    uint32_t *table;                    // [n_rasters][tiles][2] = offset, size
    unsigned long long *cursor;
    uint32_t *hist;                     // [n_tiles][kHistWords]   (pass A -> B)
    uint8_t *books;                     // [n_tiles][kBookBytes]   (pass B -> C)
    uint32_t W, rows, across, down, n_tiles;
    unsigned long long arena_cap;
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

// ------------------------------------------------------------------------
// shared by passes A and C: the tile in LDS and the row parser
// ------------------------------------------------------------------------

// One wave reads one 256-byte tile row per step (coalesced); rows need not be
// dword aligned (W = 36001 blocks), gfx950 serves unaligned dword loads.  Pixels
// outside the raster are zero, as GDAL pads edge blocks.
__device__ __forceinline__ void load_tile(const TileJob &job, const uint8_t *src, uint32_t tx, uint32_t ty,
                                          uint8_t *tile, int t)
{
    typedef uint32_t u32_u __attribute__((aligned(1)));
    const uint32_t x = tx * kTile + (uint32_t)(t & 63) * 4u;
    const uint32_t y0 = ty * kTile + (uint32_t)(t >> 6);
    uint32_t *dst = reinterpret_cast<uint32_t *>(tile) + (t & 63);
    if ((tx + 1) * kTile <= job.W && (ty + 1) * kTile <= job.rows) {
        // interior tile: 64 independent loads per thread, all in flight at once
        const uint8_t *p = src + (size_t)y0 * job.W + x;
        const size_t step = (size_t)job.W * 4;
        uint32_t v[kTile / 4];
#pragma unroll
        for (int i = 0; i < kTile / 4; i++)
            v[i] = *reinterpret_cast<const u32_u *>(p + (size_t)i * step);
#pragma unroll
        for (int i = 0; i < kTile / 4; i++)
            dst[(i * 4 + (t >> 6)) * (kRowStride / 4)] = v[i];
        return;
    }
    for (int i = 0; i < kTile / 4; i++) {
        const int r = i * 4 + (t >> 6);
        const uint32_t y = ty * kTile + (uint32_t)r;
        uint32_t v = 0;
        if (y < job.rows && x < job.W) {
            const uint8_t *p = src + (size_t)y * job.W + x;
            if (x + 4u <= job.W) {
                v = *reinterpret_cast<const u32_u *>(p);
            }
            else {
                for (uint32_t k = 0; x + k < job.W; k++)
                    v |= (uint32_t)p[k] << (8 * k);
            }
        }
        dst[r * (kRowStride / 4)] = v;
    }
}

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

// What a row parse does with each token.
enum { kCount = 0, kMeasure = 1, kEmit = 2 };

struct CodeView {               // the code book as the parser sees it (LDS)
    const uint8_t *lit_len;
    const uint16_t *lit_code;
    uint8_t dist_len[2];
    uint16_t dist_code[2];
};

struct RowEmitter {
    uint32_t *out;
    uint32_t pos;
    unsigned long long acc;
    int nacc;
    __device__ __forceinline__ void put(uint32_t value, int nbits)
    {
        acc |= (unsigned long long)value << nacc;
        nacc += nbits;
        if (nacc >= 32) {
            const uint32_t lo = (uint32_t)acc;
            const uint32_t w = pos >> 5, sh = pos & 31;
            atomicOr(&out[w], lo << sh);
            if (sh)
                atomicOr(&out[w + 1], lo >> (32 - sh));
            acc >>= 32;
            nacc -= 32;
            pos += 32;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc > 0) {
            const uint32_t lo = (uint32_t)acc & (0xffffffffu >> (32 - nacc));
            const uint32_t w = pos >> 5, sh = pos & 31;
            atomicOr(&out[w], lo << sh);
            if (sh && sh + nacc > 32)
                atomicOr(&out[w + 1], lo >> (32 - sh));
            pos += nacc;
            nacc = 0;
            acc = 0;
        }
    }
};

// Greedy parse of tile row t (matches of length >= 3 at distance 1 or 256, never
// crossing the row's end).  Literal stretches are skipped with the masks; only
// the literal bytes themselves are read from LDS.
template <int MODE>
__device__ __forceinline__ uint32_t parse_row(const uint8_t *tile, int t, const RowMasks &m,
                                              uint32_t *lit_hist, uint32_t *dist_hist,
                                              const CodeView *cv, RowEmitter *em)
{
    const uint8_t *row = tile + t * kRowStride;
    uint32_t bits = 0;
    int x = 0;

    while (x < kTile) {
        const int cand = next_candidate(m, x);
        // literals up to the next position where a match could start
        for (; x < cand; x++) {
            const int s = row[x];
            if (MODE == kCount)
                atomicAdd(&lit_hist[s], 1u);
            else if (MODE == kMeasure)
                bits += cv->lit_len[s];
            else
                em->put(cv->lit_code[s], cv->lit_len[s]);
        }
        if (x >= kTile)
            break;
        const int l1 = run_from(m.near_, x);
        const int l256 = run_from(m.far_, x);
        const bool far = l256 > l1;                 // tie: distance 1 (no extra bits)
        const int len = far ? l256 : l1;            // <= 256 - x by construction
        if (len >= 3) {
            const int lc = length_code(len);
            const int ls = 257 + lc;
            const int di = far ? 1 : 0;
            if (MODE == kCount) {
                atomicAdd(&lit_hist[ls], 1u);
                atomicAdd(&dist_hist[di], 1u);
            }
            else if (MODE == kMeasure) {
                bits += cv->lit_len[ls] + kLenExtra[lc] + cv->dist_len[di] + (far ? 6 : 0);
            }
            else {
                em->put(cv->lit_code[ls], cv->lit_len[ls]);
                if (kLenExtra[lc])
                    em->put((uint32_t)(len - kLenBase[lc]), kLenExtra[lc]);
                em->put(cv->dist_code[di], cv->dist_len[di]);
                if (far)
                    em->put(63u, 6);                // 256 - 193: distance code 15 has 6 extra bits
            }
            x += len;
        }
        else {
            const int s = row[x];
            if (MODE == kCount)
                atomicAdd(&lit_hist[s], 1u);
            else if (MODE == kMeasure)
                bits += cv->lit_len[s];
            else
                em->put(cv->lit_code[s], cv->lit_len[s]);
            x++;
        }
    }
    return bits;
}

// ------------------------------------------------------------------------
// pass A: symbol statistics + Adler-32, one workgroup per tile
// ------------------------------------------------------------------------
struct SharedA {
    uint8_t tile[kTile * kRowStride];
    uint32_t lit_hist[288];
    uint32_t dist_hist[2];
    uint32_t adler_a[kTile];
    uint32_t adler_b[kTile];
};

__global__ __launch_bounds__(kTile) void deflate_stats_kernel(const TileJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedA &sh = *reinterpret_cast<SharedA *>(smem);
    const int t = threadIdx.x;
    const uint32_t tiles = job.across * job.down;
    const uint32_t raster = blockIdx.x / tiles;
    const uint32_t tix = blockIdx.x - raster * tiles;
    const uint32_t ty = tix / job.across, tx = tix - ty * job.across;

    load_tile(job, job.rasters[raster], tx, ty, sh.tile, t);
    for (int i = t; i < 288; i += kTile)
        sh.lit_hist[i] = 0;
    if (t < 2)
        sh.dist_hist[t] = 0;
    __syncthreads();

    // Adler-32 partial sums of row t: A = sum x, B = sum (n - i) x_i over the row,
    // n = 65536, i = 256 t + k the byte's position in the tile
    {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(sh.tile + t * kRowStride);
        uint32_t a = 0, b = 0;
#pragma unroll 8
        for (int j = 0; j < kTile / 4; j++) {
            const uint32_t v = row[j];
            const uint32_t b0 = v & 0xff, b1 = (v >> 8) & 0xff, b2 = (v >> 16) & 0xff, b3 = v >> 24;
            a += b0 + b1 + b2 + b3;
            b += (uint32_t)(kTile - 4 * j) * b0 + (uint32_t)(kTile - 4 * j - 1) * b1 +
                 (uint32_t)(kTile - 4 * j - 2) * b2 + (uint32_t)(kTile - 4 * j - 3) * b3;
        }
        sh.adler_a[t] = a;      // <= 256 * 255
        sh.adler_b[t] = (uint32_t)(((unsigned long long)b +
                                    (unsigned long long)(kTileBytes - kTile * (t + 1)) * a) % 65521ull);
    }

    RowMasks m;
    row_masks(sh.tile, t, m);
    parse_row<kCount>(sh.tile, t, m, sh.lit_hist, sh.dist_hist, nullptr, nullptr);
    // a tile of one value (ocean, no-data: most of the globe): every byte repeats its
    // predecessor.  Pass C then emits it without loading the tile again.
    const bool row_constant = (m.near_[0] | (t == 0 ? 1ull : 0ull)) == ~0ull && m.near_[1] == ~0ull &&
                              m.near_[2] == ~0ull && m.near_[3] == ~0ull;
    const bool tile_uniform = __syncthreads_and(row_constant) != 0;

    uint32_t *out = job.hist + (size_t)blockIdx.x * kHistWords;
    for (int i = t; i < 288; i += kTile)
        out[i] = i == 256 ? 1u : sh.lit_hist[i];       // 256 = end of block, once
    if (t < 2)
        out[288 + t] = sh.dist_hist[t];
    // tree reduction of the 256 partial sums (both stay below 2^32)
    for (int off = kTile / 2; off > 0; off >>= 1) {
        if (t < off) {
            sh.adler_a[t] += sh.adler_a[t + off];
            sh.adler_b[t] += sh.adler_b[t + off];
        }
        __syncthreads();
    }
    if (t == 0) {
        const uint32_t s1 = (1u + sh.adler_a[0]) % 65521u;
        const uint32_t s2 = (uint32_t)(((unsigned long long)kTileBytes + sh.adler_b[0]) % 65521ull);
        out[290] = (s2 << 16) | s1;
        out[291] = tile_uniform ? (1u | ((uint32_t)sh.tile[0] << 8)) : 0u;
    }
}

// ------------------------------------------------------------------------
// pass B: code construction, one THREAD per tile (the algorithm is serial; ten
// thousand tiles per strip supply the parallelism).  Each thread works in its
// own LDS slice.
// ------------------------------------------------------------------------
// A tile with more than kMaxLive live literal/length symbols (noise, not a CN
// raster: a lookup table has at most 45 values) gets a flat code instead of a
// Huffman tree -- every live symbol ceil(log2 m) or one bit fewer, still a
// complete prefix code, matches still pay.  That bounds the tree workspace so
// 48 tiles are built per CU at a time.
constexpr int kMaxLive = 144;
struct Work {                   // Huffman workspace of one tile
    uint32_t weight[2 * kMaxLive];
    uint16_t parent[2 * kMaxLive];
    uint16_t sym[kMaxLive];
    uint8_t depth[2 * kMaxLive];
    uint8_t lit_len[288];
    uint8_t dist_len[32];
    uint16_t count[18];         // codes per length
    uint16_t next_code[18];
    uint32_t pad;               // odd dword stride: threads of a wave hit different banks
};
constexpr int kBuildThreads = 48;

// Length-limited canonical Huffman code of hist[0..n) -> len[], code[] (bit-reversed).
// Two-queue construction on symbols sorted by frequency, then the Kraft fix-up
// zlib-style encoders use to cap the depth at max_len.
__device__ void build_code(Work &w, const uint32_t *hist, int n, int max_len, uint8_t *len,
                           uint16_t *code_out)
{
    // live symbols, in symbol order; the global reads are independent of the
    // bookkeeping, eight are kept in flight
    int m = 0;
    for (int s0 = 0; s0 < n; s0 += 8) {
        uint32_t f[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            f[k] = s0 + k < n ? hist[s0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (s0 + k < n) {
                len[s0 + k] = 0;
                if (f[k]) {
                    if (m < kMaxLive) {
                        w.weight[m] = f[k];
                        w.sym[m] = (uint16_t)(s0 + k);
                    }
                    m++;
                }
            }
        }
    }
    for (int i = 0; i <= 16; i++)
        w.count[i] = 0;
    if (m > kMaxLive) {
        // flat code: 2^L - m symbols of L-1 bits, the others L bits (Kraft sum exactly 1)
        int L = 1;
        while ((1 << L) < m)
            L++;
        const int n_short = (1 << L) - m;
        int k_live = 0;
        for (int s = 0; s < n; s++) {
            if (hist[s]) {
                len[s] = (uint8_t)(k_live < n_short ? L - 1 : L);
                k_live++;
            }
        }
        w.count[L - 1] = (uint16_t)n_short;
        w.count[L] = (uint16_t)(m - n_short);
        m = 0;                      // skip the tree
    }
    // insertion sort by frequency (stable in symbol order)
    for (int i = 1; i < m; i++) {
        const uint32_t f = w.weight[i];
        const uint16_t sy = w.sym[i];
        int j = i;
        while (j > 0 && w.weight[j - 1] > f) {
            w.weight[j] = w.weight[j - 1];
            w.sym[j] = w.sym[j - 1];
            j--;
        }
        w.weight[j] = f;
        w.sym[j] = sy;
    }
    if (m == 1) {
        len[w.sym[0]] = 1;          // a lone symbol still needs one bit
        w.count[1] = 1;
    }
    if (m >= 2) {
        // leaves 0..m-1 (sorted), internal nodes m..2m-2 in non-decreasing weight order
        int leaf = 0, inode = m, next = m;
        for (; next < 2 * m - 1; next++) {
            uint32_t sum = 0;
            for (int k = 0; k < 2; k++) {
                int pick_;
                if (leaf < m && (inode >= next || w.weight[leaf] <= w.weight[inode]))
                    pick_ = leaf++;
                else
                    pick_ = inode++;
                sum += w.weight[pick_];
                w.parent[pick_] = (uint16_t)next;
            }
            w.weight[next] = sum;
        }
        w.depth[2 * m - 2] = 0;
        for (int i = 2 * m - 3; i >= 0; i--) {
            int d = w.depth[w.parent[i]] + 1;
            if (d > 64)
                d = 64;
            w.depth[i] = (uint8_t)d;
            if (i < m)
                w.count[d > max_len ? max_len : d]++;
        }
        {
            unsigned long long total = 0;
            for (int i = 1; i <= max_len; i++)
                total += (unsigned long long)w.count[i] << (max_len - i);
            while (total > (1ull << max_len)) {
                w.count[max_len]--;
                for (int i = max_len - 1; i > 0; i--) {
                    if (w.count[i]) {
                        w.count[i]--;
                        w.count[i + 1] += 2;
                        break;
                    }
                }
                total--;
            }
        }
        int idx = m - 1;            // most frequent symbols get the shortest codes
        for (int l = 1; l <= max_len; l++)
            for (int c = 0; c < w.count[l]; c++)
                len[w.sym[idx--]] = (uint8_t)l;
    }
    if (code_out) {
        // canonical codes, RFC 1951 3.2.2; count[] already holds the codes per length
        uint32_t c = 0;
        w.count[0] = 0;
        for (int l = 1; l <= max_len; l++) {
            c = (c + w.count[l - 1]) << 1;
            w.next_code[l] = (uint16_t)c;
        }
        for (int s = 0; s < n; s++) {
            const int l = len[s];
            code_out[s] = l ? (uint16_t)bitrev(w.next_code[l]++, l) : (uint16_t)0;
        }
    }
}

// the fixed canonical code of the code-length alphabet (see kClLen), bit-reversed
__device__ __forceinline__ uint32_t cl_code_of(int v)
{
    if (v < 10)
        return bitrev((uint32_t)v, 4);              // 0..9   -> 0000 .. 1001
    if (v >= 16)
        return bitrev((uint32_t)(v - 6), 4);        // 16..18 -> 1010 .. 1100
    return bitrev((uint32_t)(16 + v), 5);           // 10..15 -> 11010 .. 11111
}

// serial bit writer for the block header
struct BitWriter {
    uint32_t *out;
    uint32_t pos;
    __device__ void put(uint32_t value, int nbits)
    {
        const uint32_t w = pos >> 5, sh = pos & 31;
        out[w] |= value << sh;
        if (sh + nbits > 32)
            out[w + 1] |= value >> (32 - sh);
        pos += nbits;
    }
};

__global__ __launch_bounds__(kBuildThreads) void deflate_codes_kernel(const TileJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tile = blockIdx.x * kBuildThreads + threadIdx.x;
    if (tile >= job.n_tiles)
        return;
    Work &w = reinterpret_cast<Work *>(smem)[threadIdx.x];
    const uint32_t *hist = job.hist + (size_t)tile * kHistWords;
    Book *book = reinterpret_cast<Book *>(job.books + (size_t)tile * kBookBytes);

    build_code(w, hist, kNumLit, 15, w.lit_len, book->lit_code);
    // the distance alphabet has at most two live symbols: codes 0 and 15
    const uint32_t n_near = hist[288], n_far = hist[289];
    for (int i = 0; i < kNumDist; i++)
        w.dist_len[i] = 0;
    w.dist_len[0] = n_near ? 1 : 0;
    w.dist_len[15] = n_far ? 1 : 0;
    book->dist_len[0] = w.dist_len[0];
    book->dist_len[1] = w.dist_len[15];
    book->dist_code[0] = 0;
    book->dist_code[1] = (uint16_t)((n_near && n_far) ? 1 : 0);    // canonical: code 0 -> "0", 15 -> "1"
    for (int s = 0; s < 288; s++)
        book->lit_len[s] = s < kNumLit ? w.lit_len[s] : (uint8_t)0;

    // ---- block header (stream bits 0..15 are the zlib header), composed in LDS: the
    // tree weights are dead by now, their storage holds the header words ----
    uint32_t *hdr = w.weight;
    for (int i = 0; i < 64; i++)
        hdr[i] = 0;
    hdr[0] = 0x78u | (0x9cu << 8);              // CMF: deflate, 32K window; FLG: check bits, level 2
    BitWriter bw{ hdr, 16 };
    int hlit = kNumLit;
    while (hlit > 257 && w.lit_len[hlit - 1] == 0)
        hlit--;
    int hdist = kNumDist;
    while (hdist > 1 && w.dist_len[hdist - 1] == 0)
        hdist--;
    bw.put(1, 1);                               // BFINAL
    bw.put(2, 2);                               // BTYPE = 10, dynamic Huffman
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put(19 - 4, 4);                          // HCLEN: all 19
    for (int i = 0; i < 19; i++)
        bw.put(kClLen[kClOrder[i]], 3);
    // run-length code the hlit + hdist lengths as one sequence (RFC 1951 3.2.7)
    const int total = hlit + hdist;
    int i = 0;
    while (i < total) {
        const int v = i < hlit ? w.lit_len[i] : w.dist_len[i - hlit];
        const int vbits = v >= 10 && v < 16 ? 5 : 4;
        int run = 1;
        while (i + run < total) {
            const int j = i + run;
            if ((j < hlit ? w.lit_len[j] : w.dist_len[j - hlit]) != v)
                break;
            run++;
        }
        int left = run;
        if (v == 0) {
            while (left >= 11) {
                const int r = left > 138 ? 138 : left;
                bw.put(cl_code_of(18), 4);
                bw.put((uint32_t)(r - 11), 7);
                left -= r;
            }
            if (left >= 3) {
                bw.put(cl_code_of(17), 4);
                bw.put((uint32_t)(left - 3), 3);
                left = 0;
            }
        }
        else {
            bw.put(cl_code_of(v), vbits);
            left--;
            while (left >= 3) {
                const int r = left > 6 ? 6 : left;
                bw.put(cl_code_of(16), 4);
                bw.put((uint32_t)(r - 3), 2);
                left -= r;
            }
        }
        while (left-- > 0)
            bw.put(cl_code_of(v), vbits);
        i += run;
    }
    for (int k = 0; k < 64; k++)
        book->header[k] = hdr[k];
    book->header_bits = bw.pos;

    // ---- the stream's exact size follows from the statistics and the code lengths; its
    // place in the arena is reserved here, so pass C neither measures the tile as a whole
    // nor waits for an atomic ----
    // (<= 65536 tokens of <= 15 + 5 + 15 + 6 bits: 32-bit arithmetic is enough)
    uint32_t bits = bw.pos;
#pragma unroll 1
    for (int sy = 0; sy < kNumLit; sy++) {
        const uint32_t f = hist[sy];
        if (f)
            bits += f * ((uint32_t)w.lit_len[sy] + (sy > 256 ? (uint32_t)kLenExtra[sy - 257] : 0u));
    }
    bits += n_near * (uint32_t)w.dist_len[0] + n_far * ((uint32_t)w.dist_len[15] + 6u);
    uint32_t bytes = (bits + 7u) / 8u + 4u;
    if (bytes > (uint32_t)kMaxStream - 64u)
        bytes = (uint32_t)kMaxStream;           // stored fallback
    const unsigned long long need = (bytes + (kSlotAlign - 1)) & ~(unsigned long long)(kSlotAlign - 1);
    const unsigned long long slot = atomicAdd(job.cursor, need);
    const bool fits = slot + need <= job.arena_cap;
    book->stream_bytes = bytes;
    book->slot = fits ? (uint32_t)slot : 0xffffffffu;
    job.table[(size_t)tile * 2] = fits ? (uint32_t)slot : 0xffffffffu;
    job.table[(size_t)tile * 2 + 1] = fits ? bytes : 0u;
}

// ------------------------------------------------------------------------
// pass B, wave-parallel form: one WAVE per tile.  Same outputs as the per-thread
// kernel above (which stays as the cross-check, option "deflate_codes" = 0): sort by
// a bitonic network in LDS, the two-queue Huffman merge on register-resident queues
// (v_readlane / v_writelane with wave-uniform indices), leaf depths by parallel
// parent walks, lengths by rank, canonical codes by ballots per length, the
// code-length run-length coding by one lane per run with a prefix sum of run sizes.
// ------------------------------------------------------------------------
constexpr int kWavesPerBlock = 4;
struct WaveWork {
    uint32_t keys[256];         // (freq << 9 | symbol), sorted ascending; <= kMaxLive live
    uint16_t parent[2 * kMaxLive];
    uint8_t len[320];           // code length per lit/len symbol (0..285), then per distance code
    uint32_t hdr[68];           // 2048 header bits at most, + the word an OR may spill into
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t uni(uint32_t v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// element idx (wave-uniform) of an array spread over three registers (lane + 64 k)
__device__ __forceinline__ uint32_t lane_array_get(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t idx)
{
    const uint32_t lane = idx & 63u;
    const uint32_t which = idx >> 6;
    return which == 0 ? __builtin_amdgcn_readlane(r0, lane)
                      : which == 1 ? __builtin_amdgcn_readlane(r1, lane) : __builtin_amdgcn_readlane(r2, lane);
}

__device__ __forceinline__ void lane_array_set(uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t idx,
                                               uint32_t v, int my_lane)
{
    const bool mine = (uint32_t)my_lane == (idx & 63u);
    const uint32_t which = idx >> 6;
    if (which == 0)
        r0 = mine ? v : r0;
    else if (which == 1)
        r1 = mine ? v : r1;
    else
        r2 = mine ? v : r2;
}

// bits and tokens of one run of `n` equal code lengths `v` (RFC 1951 3.2.7), in the fixed
// code of the code-length alphabet; with a writer the tokens are emitted as well
__device__ __forceinline__ uint32_t rle_run(int v, int n, uint32_t *out, uint32_t pos)
{
    const int vbits = v >= 10 && v < 16 ? 5 : 4;
    uint32_t bits = 0;
    auto put = [&](uint32_t value, int nbits) {
        if (out) {
            const uint32_t at = pos + bits;
            const uint32_t w = at >> 5, sh = at & 31;
            atomicOr(&out[w], value << sh);
            if (sh + nbits > 32)
                atomicOr(&out[w + 1], value >> (32 - sh));
        }
        bits += nbits;
    };
    int left = n;
    if (v == 0) {
        while (left >= 11) {
            const int r = left > 138 ? 138 : left;
            put(cl_code_of(18), 4);
            put((uint32_t)(r - 11), 7);
            left -= r;
        }
        if (left >= 3) {
            put(cl_code_of(17), 4);
            put((uint32_t)(left - 3), 3);
            left = 0;
        }
    }
    else {
        put(cl_code_of(v), vbits);
        left--;
        while (left >= 3) {
            const int r = left > 6 ? 6 : left;
            put(cl_code_of(16), 4);
            put((uint32_t)(r - 3), 2);
            left -= r;
        }
    }
    while (left-- > 0)
        put(cl_code_of(v), vbits);
    return bits;
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void deflate_codes_wave_kernel(const TileJob job)
{
    __shared__ __attribute__((aligned(16))) WaveWork work[kWavesPerBlock];
    const int lane = threadIdx.x & 63;
    const uint32_t tile = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= job.n_tiles)
        return;
    WaveWork &w = work[threadIdx.x >> 6];
    const uint32_t *hist = job.hist + (size_t)tile * kHistWords;
    Book *book = reinterpret_cast<Book *>(job.books + (size_t)tile * kBookBytes);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ---- statistics: lane holds symbols lane + 64 c ----
    uint32_t h[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const int s = c * 64 + lane;
        h[c] = s < kNumLit ? hist[s] : 0u;
        w.len[s] = 0;
    }
    const uint32_t n_near = hist[288], n_far = hist[289];

    // ---- live symbols, compacted in symbol order ----
    uint32_t m = 0;
    uint32_t pos[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const bool live = h[c] != 0;
        const unsigned long long mask = __ballot(live);
        pos[c] = m + (uint32_t)__popcll(mask & lt_mask);
        if (live && pos[c] < 256u)
            w.keys[pos[c]] = (h[c] << 9) | (uint32_t)(c * 64 + lane);
        m += (uint32_t)__popcll(mask);
    }
    m = uni(m);

    uint32_t count[17];
#pragma unroll
    for (int i = 0; i <= 16; i++)
        count[i] = 0;

    if (m > (uint32_t)kMaxLive) {
        // flat code: 2^L - m symbols of L-1 bits, the others L bits
        uint32_t L = 1;
        while ((1u << L) < m)
            L++;
        const uint32_t n_short = (1u << L) - m;
#pragma unroll
        for (int c = 0; c < 5; c++)
            if (h[c])
                w.len[c * 64 + lane] = (uint8_t)(pos[c] < n_short ? L - 1 : L);
#pragma unroll
        for (int i = 1; i <= 15; i++)
            count[i] = (uint32_t)i == L - 1 ? n_short : ((uint32_t)i == L ? m - n_short : 0u);
    }
    else if (m == 1) {
        if (lane == 0)
            w.len[w.keys[0] & 0x1ffu] = 1;
        count[1] = 1;
    }
    else {
        // ---- bitonic sort of keys[0..P), P = power of two >= m, padded with the maximum ----
        uint32_t P = 2;
        while (P < m)
            P <<= 1;
        for (uint32_t i = m + lane; i < P; i += 64)
            w.keys[i] = 0xffffffffu;
        wave_sync();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < P; i += 64) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = w.keys[i], b = w.keys[partner];
                        const bool ascending = (i & k) == 0;
                        if ((a > b) == ascending) {
                            w.keys[i] = b;
                            w.keys[partner] = a;
                        }
                    }
                }
                wave_sync();
            }
        }
        // ---- two-queue Huffman merge; both queues live in registers ----
        uint32_t lw0 = lane < (int)m ? w.keys[lane] >> 9 : 0u;
        uint32_t lw1 = lane + 64 < (int)m ? w.keys[lane + 64] >> 9 : 0u;
        uint32_t lw2 = lane + 128 < (int)m ? w.keys[lane + 128] >> 9 : 0u;
        uint32_t iw0 = 0, iw1 = 0, iw2 = 0;     // internal node k (k = 0 .. m-2), created in weight order
        uint32_t leaf = 0, inode = 0;
        for (uint32_t made = 0; made + 1 < m; made++) {
            uint32_t sum = 0;
#pragma unroll
            for (int two = 0; two < 2; two++) {
                const bool have_leaf = leaf < m, have_node = inode < made;
                const uint32_t wl = have_leaf ? lane_array_get(lw0, lw1, lw2, leaf) : 0xffffffffu;
                const uint32_t wn = have_node ? lane_array_get(iw0, iw1, iw2, inode) : 0xffffffffu;
                if (have_leaf && (!have_node || wl <= wn)) {
                    if (lane == 0)
                        w.parent[leaf] = (uint16_t)(m + made);
                    sum += wl;
                    leaf++;
                }
                else {
                    if (lane == 0)
                        w.parent[m + inode] = (uint16_t)(m + made);
                    sum += wn;
                    inode++;
                }
            }
            lane_array_set(iw0, iw1, iw2, made, sum, lane);
        }
        wave_sync();
        // ---- leaf depths: every lane walks up from its leaves to the root ----
        const uint32_t root = 2 * m - 2;
        uint32_t depth[3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const uint32_t i = (uint32_t)(g * 64 + lane);
            uint32_t d = 0;
            if (i < m) {
                uint32_t p = i;
                while (p != root) {
                    p = w.parent[p];
                    d++;
                }
                if (d > 15)
                    d = 15;
            }
            depth[g] = d;
        }
#pragma unroll
        for (int L = 1; L <= 15; L++) {
            uint32_t n = 0;
#pragma unroll
            for (int g = 0; g < 3; g++)
                n += (uint32_t)__popcll(__ballot(depth[g] == (uint32_t)L && (uint32_t)(g * 64 + lane) < m));
            count[L] = n;
        }
        // ---- cap at 15 bits: restore the Kraft equality (wave-uniform scalar loop) ----
        {
            uint32_t total = 0;
#pragma unroll
            for (int L = 1; L <= 15; L++)
                total += count[L] << (15 - L);
            while (total > (1u << 15)) {
                count[15]--;
#pragma unroll
                for (int L = 14; L > 0; L--) {
                    if (count[L]) {
                        count[L]--;
                        count[L + 1] += 2;
                        break;
                    }
                }
                total--;
            }
        }
        // ---- lengths by rank: the most frequent symbol (last in the sorted list) gets the shortest ----
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const uint32_t i = (uint32_t)(g * 64 + lane);
            if (i < m) {
                const uint32_t from_top = m - 1 - i;
                uint32_t cum = 0, L = 0;
#pragma unroll
                for (int k = 1; k <= 15; k++) {
                    cum += count[k];
                    if (L == 0 && from_top < cum)
                        L = (uint32_t)k;
                }
                w.len[w.keys[i] & 0x1ffu] = (uint8_t)L;
            }
        }
    }
    // distance alphabet: codes 0 (distance 1) and 15 (distance 256)
    if (lane < 32)
        w.len[288 + lane] = 0;
    wave_sync();
    const uint32_t dl0 = n_near ? 1u : 0u, dl15 = n_far ? 1u : 0u;
    if (lane == 0) {
        w.len[288] = (uint8_t)dl0;
        w.len[288 + 15] = (uint8_t)dl15;
        book->dist_len[0] = (uint8_t)dl0;
        book->dist_len[1] = (uint8_t)dl15;
        book->dist_code[0] = 0;
        book->dist_code[1] = (uint16_t)((n_near && n_far) ? 1 : 0);
    }
    wave_sync();

    // ---- canonical codes (RFC 1951 3.2.2) and the stream's body size ----
    uint32_t next_code[16];
    {
        uint32_t c = 0;
        next_code[0] = 0;
#pragma unroll
        for (int L = 1; L <= 15; L++) {
            c = (c + (L == 1 ? 0u : count[L - 1])) << 1;
            next_code[L] = c;
        }
    }
    uint32_t body_bits = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const int s = c * 64 + lane;
        const uint32_t L = s < 288 ? w.len[s] : 0u;
        uint32_t code = 0;
#pragma unroll
        for (int k = 1; k <= 15; k++) {
            const unsigned long long mask = __ballot(L == (uint32_t)k);
            if (L == (uint32_t)k)
                code = bitrev(next_code[k] + (uint32_t)__popcll(mask & lt_mask), k);
            next_code[k] += (uint32_t)__popcll(mask);
        }
        if (s < 288) {
            book->lit_len[s] = (uint8_t)L;
            book->lit_code[s] = (uint16_t)code;
        }
        if (s < kNumLit && h[c])
            body_bits += h[c] * (L + (s > 256 ? (uint32_t)kLenExtra[s - 257] : 0u));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        body_bits += __shfl_xor(body_bits, off, 64);
    body_bits += n_near * dl0 + n_far * (dl15 + 6u);

    // ---- block header: fixed part by lane 0, the code lengths run-length coded in parallel ----
    for (int i = lane; i < 64; i += 64)
        w.hdr[i] = 0;
    // hlit / hdist: trailing zero lengths are not sent
    uint32_t hlit = 257;
#pragma unroll
    for (int c = 4; c < 5; c++) {
        const int s = c * 64 + lane;            // symbols 256 .. 319
        const unsigned long long mask = __ballot(s < kNumLit && w.len[s] != 0);
        if (mask)
            hlit = 256u + 64u - (uint32_t)__builtin_clzll(mask);     // highest live symbol + 1
        if (hlit < 257u)
            hlit = 257u;
    }
    hlit = uni(hlit);
    const uint32_t hdist = n_far ? 16u : 1u;
    const uint32_t total = hlit + hdist;
    wave_sync();
    if (lane == 0) {
        BitWriter bw{ w.hdr, 16 };
        w.hdr[0] = 0x78u | (0x9cu << 8);        // CMF: deflate, 32K window; FLG: check bits, level 2
        bw.put(1, 1);                           // BFINAL
        bw.put(2, 2);                           // BTYPE = 10, dynamic Huffman
        bw.put(hlit - 257u, 5);
        bw.put(hdist - 1u, 5);
        bw.put(19 - 4, 4);                      // HCLEN: all 19
        for (int i = 0; i < 19; i++)
            bw.put(kClLen[kClOrder[i]], 3);
    }
    wave_sync();
    const uint32_t fixed_bits = 16 + 3 + 5 + 5 + 4 + 19 * 3;
    // the sequence of code lengths: lit/len 0..hlit-1, then distance codes 0..hdist-1
    auto seq = [&](uint32_t i) -> int { return i < hlit ? w.len[i] : w.len[288 + (i - hlit)]; };
    unsigned long long start_mask[5];
    int v_at[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const uint32_t i = (uint32_t)(c * 64 + lane);
        const bool in = i < total;
        v_at[c] = in ? seq(i) : -1;
        const bool start = in && (i == 0 || seq(i - 1) != v_at[c]);
        start_mask[c] = __ballot(start);
    }
    uint32_t run_pos = fixed_bits;              // wave-uniform: bits before this chunk's runs
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const uint32_t i = (uint32_t)(c * 64 + lane);
        const bool start = (start_mask[c] >> lane) & 1ull;
        uint32_t run_len = 0;
        if (start) {
            // next run start after i, in this chunk or a later one
            uint32_t nxt = total;
            const unsigned long long rest = lane < 63 ? start_mask[c] >> (lane + 1) : 0ull;
            if (rest) {
                nxt = i + 1u + (uint32_t)__builtin_ctzll(rest);
            }
            else {
#pragma unroll
                for (int c2 = 4; c2 > 0; c2--)      // keep the nearest later chunk with a start
                    if (c2 > c && start_mask[c2])
                        nxt = (uint32_t)(c2 * 64) + (uint32_t)__builtin_ctzll(start_mask[c2]);
            }
            run_len = nxt - i;
        }
        const uint32_t bits = start ? rle_run(v_at[c], (int)run_len, nullptr, 0) : 0u;
        uint32_t incl = bits;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += up;
        }
        if (start)
            rle_run(v_at[c], (int)run_len, w.hdr, run_pos + incl - bits);
        run_pos += uni(__shfl(incl, 63, 64));
    }
    wave_sync();
    const uint32_t header_bits = run_pos;
    book->header[lane] = w.hdr[lane];

    if (lane == 0) {
        const uint32_t bits_total = header_bits + body_bits;
        uint32_t bytes = (bits_total + 7u) / 8u + 4u;
        if (bytes > (uint32_t)kMaxStream - 64u)
            bytes = (uint32_t)kMaxStream;       // stored fallback
        const unsigned long long need = (bytes + (kSlotAlign - 1)) & ~(unsigned long long)(kSlotAlign - 1);
        const unsigned long long slot = atomicAdd(job.cursor, need);
        const bool fits = slot + need <= job.arena_cap;
        book->header_bits = header_bits;
        book->stream_bytes = bytes;
        book->slot = fits ? (uint32_t)slot : 0xffffffffu;
        job.table[(size_t)tile * 2] = fits ? (uint32_t)slot : 0xffffffffu;
        job.table[(size_t)tile * 2 + 1] = fits ? bytes : 0u;
    }
}

// ------------------------------------------------------------------------
// pass C: measure, place and emit; one workgroup per tile
// ------------------------------------------------------------------------
template <bool SMALL>
struct SharedC {
    static constexpr int kWords = SMALL ? kSmallStream / 4 + 16 : kOutWords;
    uint8_t tile[kTile * kRowStride];
    uint32_t out[kWords];
    uint8_t lit_len[288];
    uint16_t lit_code[288];
    uint32_t wave_sum[4];
};

// inclusive prefix sum over the 256 threads of the workgroup (4 waves of 64)
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *wave_sum, int t)
{
    const int lane = t & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off, 64);
        if (lane >= off)
            v += up;
    }

    if (lane == 63)
        wave_sum[t >> 6] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (t >> 6); w++)
        base += wave_sum[w];
    return v + base;
}

// SMALL = true : streams of at most kSmallStream bytes, LDS for two workgroups per CU
// SMALL = false: the others (noisy tiles, stored fallback), one workgroup per CU
template <bool SMALL>
__global__ __launch_bounds__(kTile) void deflate_emit_kernel(const TileJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SharedC<SMALL> &sh = *reinterpret_cast<SharedC<SMALL> *>(smem);
    const int t = threadIdx.x;
    const Book *book = reinterpret_cast<const Book *>(job.books + (size_t)blockIdx.x * kBookBytes);
    const uint32_t stream_bytes = book->stream_bytes;
    if ((stream_bytes <= (uint32_t)kSmallStream) != SMALL)
        return;                                     // the other launch emits this tile
    const uint32_t slot = book->slot;
    if (slot == 0xffffffffu)
        return;                                     // arena too small: the table says so
    const uint32_t tiles = job.across * job.down;
    const uint32_t raster = blockIdx.x / tiles;
    const uint32_t tix = blockIdx.x - raster * tiles;
    const uint32_t ty = tix / job.across, tx = tix - ty * job.across;
    const uint32_t header_bits = book->header_bits;
    const uint32_t adler = job.hist[(size_t)blockIdx.x * kHistWords + 290];
    const bool stored = stream_bytes == (uint32_t)kMaxStream;
    const uint32_t n_words = (stream_bytes + 3) / 4;
    const uint32_t uniform_word = job.hist[(size_t)blockIdx.x * kHistWords + 291];
    const bool uniform = (uniform_word & 1u) != 0 && !stored;     // one-value tile: no need for its bytes

    if (!uniform)
        load_tile(job, job.rasters[raster], tx, ty, sh.tile, t);
    // the output image: block header from pass B, zeros up to the stream's end
    for (uint32_t i = t; i < n_words + 1; i += kTile)
        sh.out[i] = (i < 64 && !stored) ? book->header[i] : 0u;
    for (int i = t; i < 288; i += kTile) {
        sh.lit_len[i] = book->lit_len[i];
        sh.lit_code[i] = book->lit_code[i];
    }
    CodeView cv;
    cv.lit_len = sh.lit_len;
    cv.lit_code = sh.lit_code;
    cv.dist_len[0] = book->dist_len[0];
    cv.dist_len[1] = book->dist_len[1];
    cv.dist_code[0] = book->dist_code[0];
    cv.dist_code[1] = book->dist_code[1];
    __syncthreads();

    uint8_t *o = reinterpret_cast<uint8_t *>(sh.out);
    if (uniform) {
        // the greedy parse of a one-value tile, written out: row 0 = literal + match(255, 1),
        // every other row = match(256, 1); both lengths use length code 27 (227..257, 5 extra bits)
        const uint32_t v = (uniform_word >> 8) & 0xffu;
        const int ls = 257 + 27;
        const uint32_t match_bits = (uint32_t)sh.lit_len[ls] + 5u + cv.dist_len[0];
        const uint32_t row0_bits = (uint32_t)sh.lit_len[v] + match_bits;
        RowEmitter em{ sh.out, header_bits + (t == 0 ? 0u : row0_bits + (uint32_t)(t - 1) * match_bits), 0ull, 0 };
        if (t == 0)
            em.put(sh.lit_code[v], sh.lit_len[v]);
        em.put(sh.lit_code[ls], sh.lit_len[ls]);
        em.put((uint32_t)((t == 0 ? 255 : 256) - 227), 5);
        em.put(cv.dist_code[0], cv.dist_len[0]);
        if (t == kTile - 1)
            em.put(sh.lit_code[256], sh.lit_len[256]);      // end of block
        em.finish();
    }
    else if (!stored) {
        RowMasks m;
        row_masks(sh.tile, t, m);
        const uint32_t bits = parse_row<kMeasure>(sh.tile, t, m, nullptr, nullptr, &cv, nullptr);
        const uint32_t incl = block_scan(bits, sh.wave_sum, t);
        RowEmitter em{ sh.out, header_bits + incl - bits, 0ull, 0 };
        parse_row<kEmit>(sh.tile, t, m, nullptr, nullptr, &cv, &em);
        if (t == kTile - 1)
            em.put(sh.lit_code[256], sh.lit_len[256]);      // end of block
        em.finish();
    }
    else if (!SMALL) {
        // stored fallback: two blocks of 32768 bytes (LEN is 16 bit)
        if (t == 0) {
            o[0] = 0x78;
            o[1] = 0x01;
            for (int b = 0; b < 2; b++) {
                uint8_t *h = o + 2 + b * (5 + 32768);
                h[0] = (uint8_t)(b == 1);       // BFINAL, BTYPE = 00, padded to the byte
                h[1] = 0x00;
                h[2] = 0x80;                    // LEN = 32768
                h[3] = 0xff;
                h[4] = 0x7f;                    // NLEN
            }
        }
        // row t = bytes 256 t .. 256 t + 255 of the tile; block b holds rows 128 b ..
        uint8_t *dst = o + 2 + (t >> 7) * (5 + 32768) + 5 + (t & 127) * kTile;
        const uint8_t *row = sh.tile + t * kRowStride;
        for (int k = 0; k < kTile; k++)
            dst[k] = row[k];
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t at = stream_bytes - 4;
        o[at] = (uint8_t)(adler >> 24);
        o[at + 1] = (uint8_t)(adler >> 16);
        o[at + 2] = (uint8_t)(adler >> 8);
        o[at + 3] = (uint8_t)adler;
    }
    __syncthreads();
    {
        const uint32_t nvec = (stream_bytes + 15) / 16;
        u32x4 *dst = reinterpret_cast<u32x4 *>(job.arena + slot);
        const u32x4 *srcv = reinterpret_cast<const u32x4 *>(sh.out);
        for (uint32_t i = t; i < nvec; i += kTile)
            dst[i] = srcv[i];
    }
}

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

size_t gcn10_gpu_deflate_arena_bound(int W, int rows, int n_rasters)
{
    if (W <= 0 || rows <= 0 || n_rasters <= 0)
        return 0;
    const size_t across = ((size_t)W + kTile - 1) / kTile, down = ((size_t)rows + kTile - 1) / kTile;
    const size_t slot = ((size_t)kMaxStream + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
    return across * down * (size_t)n_rasters * slot;
}

int gcn10_gpu_deflate_strip(gcn10_gpu_ctx *ctx, const uint8_t *const *rasters_dev, int n_rasters, int W,
                            int rows, uint8_t *arena_dev, size_t arena_cap, uint32_t *table_dev,
                            unsigned long long *cursor_dev, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (n_rasters < 1 || n_rasters > GCN10_N_RASTERS || W <= 0 || rows < 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_strip: bad shape %d rasters of %d x %d", n_rasters, W, rows);
    if (rows == 0)
        return GCN10_OK;
    if (!rasters_dev || !arena_dev || !table_dev || !cursor_dev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_strip: null pointer");
    if ((reinterpret_cast<uintptr_t>(arena_dev) & 15u) != 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_strip: arena must be 16-byte aligned");
    TileJob job;
    job.rasters = rasters_dev;
    job.arena = arena_dev;
    job.table = table_dev;
    job.cursor = cursor_dev;
    job.W = (uint32_t)W;
    job.rows = (uint32_t)rows;
    job.across = ((uint32_t)W + kTile - 1) / kTile;
    job.down = ((uint32_t)rows + kTile - 1) / kTile;
    job.arena_cap = arena_cap;
    const uint64_t nblocks = (uint64_t)job.across * job.down * (uint64_t)n_rasters;
    if (nblocks > 0x7fffffffull)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_strip: too many tiles");
    job.n_tiles = (uint32_t)nblocks;

    // per-tile statistics and code books: workspace owned by the context
    const size_t need = (size_t)nblocks * ((size_t)kHistWords * 4 + (size_t)kBookBytes);
    if (need > ctx->deflate_ws_cap) {
        HIP_TRY(hipDeviceSynchronize());        // the old workspace may still be in use
        if (ctx->deflate_ws)
            HIP_TRY(hipFree(ctx->deflate_ws));
        ctx->deflate_ws = nullptr;
        ctx->deflate_ws_cap = 0;
        HIP_TRY(hipMalloc(&ctx->deflate_ws, need));
        ctx->deflate_ws_cap = need;
    }
    job.hist = reinterpret_cast<uint32_t *>(ctx->deflate_ws);
    job.books = reinterpret_cast<uint8_t *>(ctx->deflate_ws) + (size_t)nblocks * kHistWords * 4;

    static_assert(sizeof(SharedC<false>) <= 160 * 1024, "tile + output image must fit the CU's 160 KiB of LDS");
    static_assert(sizeof(SharedC<true>) <= 80 * 1024, "two small-stream workgroups must fit one CU");
    static_assert(sizeof(Work) * kBuildThreads <= 160 * 1024, "code construction slices must fit LDS");
    static_assert(kBookBytes % 4 == 0, "code books are dword aligned");
    if (!ctx->deflate_ready) {
        // more than 64 KiB of dynamic LDS has to be asked for, once per device
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_stats_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedA)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_codes_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(Work) * kBuildThreads)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_emit_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedC<true>)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_emit_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedC<false>)));
        ctx->deflate_ready = true;
    }
    hipStream_t s = as_stream(ctx, stream);
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(deflate_stats_kernel, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedA), s, job);
    if (ctx->deflate_wave_codes)
        hipLaunchKernelGGL(deflate_codes_wave_kernel,
                           dim3(((uint32_t)nblocks + kWavesPerBlock - 1) / kWavesPerBlock),
                           dim3(64 * kWavesPerBlock), 0, s, job);
    else
        hipLaunchKernelGGL(deflate_codes_kernel, dim3(((uint32_t)nblocks + kBuildThreads - 1) / kBuildThreads),
                           dim3(kBuildThreads), sizeof(Work) * kBuildThreads, s, job);
    hipLaunchKernelGGL(deflate_emit_kernel<true>, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedC<true>), s,
                       job);
    hipLaunchKernelGGL(deflate_emit_kernel<false>, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedC<false>),
                       s, job);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

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
    if (need > ctx->deflate_ws_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (ctx->deflate_ws)
            HIP_TRY(hipFree(ctx->deflate_ws));
        ctx->deflate_ws = nullptr;
        ctx->deflate_ws_cap = 0;
        HIP_TRY(hipMalloc(&ctx->deflate_ws, need));
        ctx->deflate_ws_cap = need;
    }
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
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_codes_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(Work) * kBuildThreads)));
        ctx->fused_ready = true;
    }
    hipStream_t s = as_stream(ctx, stream);
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(fused_stats_kernel, dim3(positions), dim3(kTile), sizeof(SharedFA), s, job);
    if (ctx->deflate_wave_codes)
        hipLaunchKernelGGL(deflate_codes_wave_kernel,
                           dim3(((uint32_t)nblocks + kWavesPerBlock - 1) / kWavesPerBlock),
                           dim3(64 * kWavesPerBlock), 0, s, job.t);
    else
        hipLaunchKernelGGL(deflate_codes_kernel, dim3(((uint32_t)nblocks + kBuildThreads - 1) / kBuildThreads),
                           dim3(kBuildThreads), sizeof(Work) * kBuildThreads, s, job.t);
    const uint32_t groups = (job.n_sel + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(fused_emit_kernel, dim3(positions, groups), dim3(kTile), sizeof(SharedFC), s, job);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

}  // extern "C"

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

// the 19 three-bit entries of the block header that describe the fixed code-length code, in kClOrder, as one number
constexpr unsigned long long cl_len_bits()
{
    constexpr uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    constexpr uint8_t len[19] = { 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 4, 4, 4 };
    unsigned long long v = 0;
    for (int i = 0; i < 19; i++)
        v |= (unsigned long long)len[order[i]] << (3 * i);
    return v;
}
constexpr unsigned long long kClLenBits = cl_len_bits();       // 57 bits

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

// placement by pass F-C (fused encoder): every stream's slot size, summed per raster and 64 positions
__device__ __forceinline__ void add_chunk_total(const TileJob &job, uint32_t tile, uint32_t bytes)
{
    if (job.chunk_tot) {
        const uint32_t per_raster = job.across * job.down;
        const uint32_t r = tile / per_raster, pos = tile - r * per_raster;
        atomicAdd(&job.chunk_tot[r * job.n_chunks + (pos >> 6)], (bytes + (uint32_t)(kSlotAlign - 1)) & ~(uint32_t)(kSlotAlign - 1));
    }
}

__global__ __launch_bounds__(kBuildThreads) void deflate_codes_kernel(const TileJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tile = blockIdx.x * kBuildThreads + threadIdx.x;
    if (tile >= job.n_tiles)
        return;
    Work &w = reinterpret_cast<Work *>(smem)[threadIdx.x];
    const uint32_t *hist = job.hist + (size_t)tile * kHistWords;
    Book *book = reinterpret_cast<Book *>(job.books + (size_t)tile * kBookBytes);
    if (hist[291] & kAliasFlag) {
        // the fused encoder found this tile equal to another raster's: nothing to build
        book->stream_bytes = 0;
        book->slot = kAliasSlot;
        job.sizes[(size_t)tile * 2] = kAliasSlot;
        job.sizes[(size_t)tile * 2 + 1] = 0u;
        return;
    }

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
    // (its place in the arena: deflate_place_kernel, the next launch, from the sizes in the table)
    book->stream_bytes = bytes;
    job.sizes[(size_t)tile * 2] = 0u;
    job.sizes[(size_t)tile * 2 + 1] = bytes;
    add_chunk_total(job, tile, bytes);
}

// ------------------------------------------------------------------------
// pass B, wave-parallel form: one WAVE per tile.  Same outputs as the per-thread
// kernel above (which stays as the cross-check, option "deflate_codes" = 0): sort by
// a bitonic network in LDS, the two-queue Huffman merge on register-resident queues
// (v_readlane / v_writelane with wave-uniform indices), leaf depths by parallel
// parent walks, lengths by rank, canonical codes by ballots per length, the
// code-length run-length coding by one lane per run with a prefix sum of run sizes.
// ------------------------------------------------------------------------
constexpr int kWavesPerBlock = 8;      // (divides 64: a workgroup's tiles lie in one chunk of the placement)
struct alignas(16) WaveWork {
    uint32_t keys[256];         // (freq << 9 | symbol), sorted ascending; <= kMaxLive live
    uint32_t hdr[68];           // 2048 header bits at most, + the word an OR may spill into (8-byte aligned: ds_or_b64)
    uint16_t parent[2 * kMaxLive];
    uint8_t len[320];           // code length per lit/len symbol (0..285), then per distance code
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

// The tokens of rle_run() for a run of `n` equal code lengths `v` as one bit string (LSB first), without loops:
// zeros take at most 2 x 11 + 11 bits; other lengths fit 64 bits up to n = kRleStringMax (the length, at most
// eight repeats of six, a rest of up to five).  Returns the number of bits.
constexpr int kRleStringMax = 54;
__device__ __forceinline__ uint32_t rle_run_string(int v, int n, unsigned long long &s)
{
    if (v == 0) {
        const uint32_t q = (uint32_t)n / 138u, rem = (uint32_t)n - q * 138u;
        const unsigned long long p11 = (unsigned long long)(cl_code_of(18) | 127u << 4);
        s = q == 0u ? 0ull : q == 1u ? p11 : (p11 | p11 << 11);
        const uint32_t nb = q * 11u;
        // rest: 11..137 -> symbol 18, 3..10 -> symbol 17, 1..2 -> that many codes of length 0 (all-zero bits)
        const unsigned long long tail = rem >= 11u ? (unsigned long long)(cl_code_of(18) | (rem - 11u) << 4)
                                                   : rem >= 3u ? (unsigned long long)(cl_code_of(17) | (rem - 3u) << 4) : 0ull;
        const uint32_t tb = rem >= 11u ? 11u : rem >= 3u ? 7u : rem * 4u;
        s |= tail << nb;
        return nb + tb;
    }
    const uint32_t vbits = v >= 10 && v < 16 ? 5u : 4u;
    const unsigned long long cv = cl_code_of(v);
    const uint32_t left = (uint32_t)n - 1u, q = left / 6u, rem = left - q * 6u;
    // q times "repeat six" (symbol 16, extra bits 3): the six-bit pattern times ones at every sixth bit
    const unsigned long long ones = 0x0041041041041041ull & ((1ull << (6u * q)) - 1ull);
    const unsigned long long rep = (unsigned long long)(cl_code_of(16) | 3u << 4) * ones;
    const unsigned long long tail = rem >= 3u ? (unsigned long long)(cl_code_of(16) | (rem - 3u) << 4)
                                              : rem == 2u ? (cv | cv << vbits) : rem == 1u ? cv : 0ull;
    const uint32_t tb = rem >= 3u ? 6u : rem * vbits;
    s = cv | rep << vbits | tail << (vbits + 6u * q);
    return vbits + 6u * q + tb;
}

// bits rle_run() spends on a run of `n` equal code lengths `v`, without walking it
__device__ __forceinline__ uint32_t rle_run_bits(int v, int n)
{
    if (v == 0) {
        // runs of 11..138 zeros: symbol 18 (4 + 7 bits); 3..10: symbol 17 (4 + 3); fewer: 4 bits each
        const int q = n / 138, rem = n - q * 138;
        return (uint32_t)(q * 11 + (rem >= 11 ? 11 : rem >= 3 ? 7 : rem * 4));
    }
    // the length itself, then repeats of 3..6 (symbol 16: 4 + 2 bits), then what is left, one by one
    const int vbits = v >= 10 && v < 16 ? 5 : 4;
    const int left = n - 1, q = left / 6, rem = left - q * 6;
    return (uint32_t)(vbits + q * 6 + (rem >= 3 ? 6 : rem * vbits));
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void deflate_codes_wave_kernel(const TileJob job)
{
    __shared__ __attribute__((aligned(16))) WaveWork work[kWavesPerBlock];
    __shared__ uint32_t wg_bytes, wg_done;
    const int lane = threadIdx.x & 63;
    // a workgroup = kWavesPerBlock consecutive tiles of ONE raster (blockIdx.y): its streams' slot bytes are added up
    // in LDS and go to the placement's chunk total (pass F-C) as one atomic per workgroup -- one per wave was 7 614
    // device-scope atomics on 126 neighbouring words per strip, which queue up in one memory channel (+8 us)
    if (threadIdx.x == 0) {
        wg_bytes = 0;
        wg_done = 0;
    }
    __syncthreads();
    const uint32_t per_raster = job.across * job.down;
    const uint32_t tile_pos = uni(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile_pos >= per_raster)
        return;
    const uint32_t tile = blockIdx.y * per_raster + tile_pos;
    const uint32_t live_waves = per_raster - blockIdx.x * kWavesPerBlock < (uint32_t)kWavesPerBlock
                                    ? per_raster - blockIdx.x * kWavesPerBlock : (uint32_t)kWavesPerBlock;
    auto add_up = [&](uint32_t bytes) {         // lane 0 of every wave that finishes its tile, once
        if (!job.chunk_tot)
            return;
        if (bytes)
            atomicAdd(&wg_bytes, (bytes + (uint32_t)(kSlotAlign - 1)) & ~(uint32_t)(kSlotAlign - 1));
        // (LDS executes a wave's operations in order: whoever counts the last wave in sees every sum before it)
        if (__hip_atomic_fetch_add(&wg_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == live_waves - 1u) {
            const uint32_t total = __hip_atomic_load(&wg_bytes, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (total)
                atomicAdd(&job.chunk_tot[blockIdx.y * job.n_chunks + (tile_pos >> 6)], total);
        }
    };
    WaveWork &w = work[threadIdx.x >> 6];
    const uint32_t *hist = job.hist + (size_t)tile * kHistWords;
    Book *book = reinterpret_cast<Book *>(job.books + (size_t)tile * kBookBytes);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (uni(hist[291]) & kAliasFlag) {
        // the fused encoder found this tile equal to another raster's: nothing to build
        if (lane == 0) {
            book->stream_bytes = 0;
            book->slot = kAliasSlot;
            job.sizes[(size_t)tile * 2] = kAliasSlot;
            job.sizes[(size_t)tile * 2 + 1] = 0u;
            add_up(0u);
        }
        return;
    }

    // (timing experiments, option "codes_stop" = p + 1: leave after phase p; books and sizes of the last complete
    // launch stay in the workspace, so the passes behind this one still see valid input)
    const uint32_t stop_after = job.codes_stop ? job.codes_stop - 1u : 99u;
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

    if (stop_after == 0u)
        return;                 // phase 0: statistics loaded, live symbols compacted
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
        if (m <= 63u) {
            // ---- the usual case: every lane ranks its own key among the m (keys are distinct: the symbol is in
            // them) with one v_readlane and a compare per key, and puts it where it belongs -- ~3 m instructions
            // instead of the 21 compare-exchange passes through LDS of the network below ----
            wave_sync();
            const uint32_t key = (uint32_t)lane < m ? w.keys[lane] : 0xffffffffu;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; j++)
                rank += (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j) < key ? 1u : 0u;
            wave_sync();
            if ((uint32_t)lane < m)
                w.keys[rank] = key;
            wave_sync();
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
        }
        if (stop_after == 1u)
            return;             // phase 1: + the sort
        // ---- two-queue Huffman merge; both queues live in registers ----
        if (m <= 63u) {
            // The usual case (a tile has a few dozen live symbols): one leaf and one internal node per lane, so
            // a pick is two v_readlane, a scalar compare and two selects -- no LDS, no branch on which register
            // holds the element.  Lanes that hold no leaf / no node yet carry an infinite weight, which makes the
            // "queue is empty" tests of the general form below unnecessary.  (Timing switches, round 3: this loop
            // was 49 of the kernel's 81 us per noisy strip in its general form.)
            uint32_t lw = (uint32_t)lane < m ? w.keys[lane] >> 9 : 0xffffffffu;
            uint32_t iw = 0xffffffffu;              // internal node k (k = 0 .. m-2), created in weight order
            uint32_t pl = 0, pn = 0;                // parent of leaf `lane`, of internal node `lane`
            // Written out (the compiler keeps the queue heads in vector registers and compares them there): the two
            // heads, the two queue positions and the sum live in scalar registers; a pick is a scalar compare and
            // branch, one v_writelane for the parent (lane select in M0: two different SGPR operands would break the
            // constant-bus rule) and one v_readlane for the new head of the queue it took from -- 5 vector
            // instructions per merge (14 in the compiler's form of the loop with selects).
            {
                uint32_t wl, wn, leaf, inode, made, sum, par, m0_saved;
                const uint32_t last = m - 1u;           // m >= 2 here
                asm volatile("s_mov_b32 %[m0s], m0\n\t"
                             "v_readlane_b32 %[wl], %[lw], 0\n\t"
                             "s_mov_b32 %[wn], -1\n\t"
                             "s_mov_b32 %[leaf], 0\n\t"
                             "s_mov_b32 %[inode], 0\n\t"
                             "s_mov_b32 %[made], 0\n"
                             "1:\n\t"
                             "s_add_u32 %[par], %[m], %[made]\n\t"
                             "s_mov_b32 %[sum], 0\n\t"
                             // first pick (tie: the leaf)
                             "s_cmp_le_u32 %[wl], %[wn]\n\t"
                             "s_cbranch_scc0 2f\n\t"
                             "s_mov_b32 m0, %[leaf]\n\t"
                             "s_add_u32 %[sum], %[sum], %[wl]\n\t"
                             "s_add_u32 %[leaf], %[leaf], 1\n\t"
                             "v_writelane_b32 %[pl], %[par], m0\n\t"
                             "v_readlane_b32 %[wl], %[lw], %[leaf]\n\t"        // (lanes m .. 63: infinite)
                             "s_branch 3f\n"
                             "2:\n\t"
                             "s_mov_b32 m0, %[inode]\n\t"
                             "s_add_u32 %[sum], %[sum], %[wn]\n\t"
                             "s_add_u32 %[inode], %[inode], 1\n\t"
                             "v_writelane_b32 %[pn], %[par], m0\n\t"
                             "v_readlane_b32 %[wn], %[iw], %[inode]\n"          // (not made yet: infinite)
                             "3:\n\t"
                             // second pick
                             "s_cmp_le_u32 %[wl], %[wn]\n\t"
                             "s_cbranch_scc0 4f\n\t"
                             "s_mov_b32 m0, %[leaf]\n\t"
                             "s_add_u32 %[sum], %[sum], %[wl]\n\t"
                             "s_add_u32 %[leaf], %[leaf], 1\n\t"
                             "v_writelane_b32 %[pl], %[par], m0\n\t"
                             "v_readlane_b32 %[wl], %[lw], %[leaf]\n\t"
                             "s_branch 5f\n"
                             "4:\n\t"
                             "s_mov_b32 m0, %[inode]\n\t"
                             "s_add_u32 %[sum], %[sum], %[wn]\n\t"
                             "s_add_u32 %[inode], %[inode], 1\n\t"
                             "v_writelane_b32 %[pn], %[par], m0\n\t"
                             "v_readlane_b32 %[wn], %[iw], %[inode]\n"
                             "5:\n\t"
                             // the new node: iw[made] = sum; it is the head of its queue if that queue was empty
                             "s_mov_b32 m0, %[made]\n\t"
                             "s_cmp_eq_u32 %[inode], %[made]\n\t"
                             "s_cselect_b32 %[wn], %[sum], %[wn]\n\t"
                             "v_writelane_b32 %[iw], %[sum], m0\n\t"
                             "s_add_u32 %[made], %[made], 1\n\t"
                             "s_cmp_lt_u32 %[made], %[last]\n\t"
                             "s_cbranch_scc1 1b\n\t"
                             "s_mov_b32 m0, %[m0s]"                              // (M0 is the compiler's: put it back)
                             : [wl] "=&s"(wl), [wn] "=&s"(wn), [leaf] "=&s"(leaf), [inode] "=&s"(inode), [made] "=&s"(made),
                               [sum] "=&s"(sum), [par] "=&s"(par), [m0s] "=&s"(m0_saved), [iw] "+v"(iw), [pl] "+v"(pl),
                               [pn] "+v"(pn)
                             : [lw] "v"(lw), [m] "s"(m), [last] "s"(last)
                             : "scc");
            }
            if ((uint32_t)lane < m)
                w.parent[lane] = (uint16_t)pl;
            if ((uint32_t)lane + 1u < m)
                w.parent[m + (uint32_t)lane] = (uint16_t)pn;
        }
        else {
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
        }
        wave_sync();
        if (stop_after == 2u)
            return;             // phase 2: + the merge
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
    if (stop_after == 3u)
        return;                 // phase 3: + depths, counts per length, the 15-bit cap, lengths by rank
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
    uint32_t max_len = 0;                       // (wave-uniform; a CN tile's codes seldom reach 8 bits)
#pragma unroll
    for (int L = 1; L <= 15; L++)
        max_len = count[L] ? (uint32_t)L : max_len;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const int s = c * 64 + lane;
        const uint32_t L = s < 288 ? w.len[s] : 0u;
        uint32_t code = 0;
        if (__ballot(L != 0u) != 0ull) {        // (CN values leave whole chunks of 64 symbols unused)
#pragma unroll
            for (int k = 1; k <= 15; k++) {
                if ((uint32_t)k <= max_len) {
                    const unsigned long long mask = __ballot(L == (uint32_t)k);
                    if (L == (uint32_t)k)
                        code = bitrev(next_code[k] + (uint32_t)__popcll(mask & lt_mask), k);
                    next_code[k] += (uint32_t)__popcll(mask);
                }
            }
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

    if (stop_after == 4u)
        return;                 // phase 4: + canonical codes, body size
    // ---- block header: fixed part by lane 0, the code lengths run-length coded in parallel ----
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
    {
        // the fixed part in closed form (a serial bit writer on lane 0 took 24 read-modify-write trips to LDS):
        // CMF/FLG (deflate, 32K window; check bits, level 2), BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN = all 19,
        // then the 19 three-bit lengths of the fixed code-length code -- a constant
        const unsigned long long lo = 0x9c78ull | 5ull << 16 | (unsigned long long)(hlit - 257u) << 19 |
                                      (unsigned long long)(hdist - 1u) << 24 | 15ull << 29 | kClLenBits << 33;
        const uint32_t hi = (uint32_t)(kClLenBits >> 31);
        w.hdr[lane] = lane == 0 ? (uint32_t)lo : lane == 1 ? (uint32_t)(lo >> 32) : lane == 2 ? hi : 0u;
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
        if (start_mask[c] == 0ull)
            continue;                           // (a chunk inside one long run of zeros)
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
        if (__ballot(start && v_at[c] != 0 && run_len > (uint32_t)kRleStringMax) == 0ull) {
            // the usual case: every run's tokens as one string of at most 63 bits, one or two 64-bit ORs into LDS
            unsigned long long str = 0;
            const uint32_t bits = start ? rle_run_string(v_at[c], (int)run_len, str) : 0u;
            const uint32_t incl = wave_scan_dpp(bits);
            if (start) {
                const uint32_t at = run_pos + incl - bits;
                unsigned long long *h64 = reinterpret_cast<unsigned long long *>(w.hdr);
                const uint32_t sh = at & 63u;
                atomicOr(&h64[at >> 6], str << sh);
                if (sh + bits > 64u)
                    atomicOr(&h64[(at >> 6) + 1u], str >> (64u - sh));
            }
            run_pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            continue;
        }
        const uint32_t bits = start ? rle_run_bits(v_at[c], (int)run_len) : 0u;
        const uint32_t incl = wave_scan_dpp(bits);
        if (start)
            rle_run(v_at[c], (int)run_len, w.hdr, run_pos + incl - bits);
        run_pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    wave_sync();
    const uint32_t header_bits = run_pos;
    book->header[lane] = w.hdr[lane];

    if (lane == 0) {
        const uint32_t bits_total = header_bits + body_bits;
        uint32_t bytes = (bits_total + 7u) / 8u + 4u;
        if (bytes > (uint32_t)kMaxStream - 64u)
            bytes = (uint32_t)kMaxStream;       // stored fallback
        // (its place in the arena: deflate_place_kernel, the next launch, from the sizes in the table)
        book->header_bits = header_bits;
        book->stream_bytes = bytes;
        job.sizes[(size_t)tile * 2] = 0u;
        job.sizes[(size_t)tile * 2 + 1] = bytes;
        add_up(bytes);
    }
}

// ------------------------------------------------------------------------
// pass B': where every stream goes.  One workgroup; streams are laid out in index order
// (raster-major, tiles of a raster row-major), 16-byte aligned, and every raster's first
// stream starts at a multiple of `seg_align`: the streams of one raster of a strip are ONE
// contiguous extent of the arena, in tile order, so the host appends them to the raster's
// GeoTIFF with a single write (and, with seg_align = 4096, may do so with O_DIRECT straight
// from the pinned copy of the arena).  Replaces the atomic slot reservation of rounds 1-2,
// whose order was the order in which the code-construction waves happened to finish.
// Aliases (fused encoder: this raster's tile is another raster's stream) take no room.
// *cursor = arena bytes used, pads included; a stream that does not fit gets offset
// 0xffffffff and size 0, as before.
// ------------------------------------------------------------------------
constexpr int kPlaceThreads = 1024;

// Pass B left table[i] = { alias mark or 0, stream bytes }: 8 contiguous bytes per stream, so a thread's
// K consecutive entries are a few cache lines (the first form of this kernel read the sizes out of the
// 1.1 KB code books: 54 us per strip; this one: a few us).
__device__ __forceinline__ uint32_t place_need(uint32_t mark, uint32_t bytes)
{
    return mark == kAliasSlot ? 0u : (bytes + (uint32_t)(kSlotAlign - 1)) & ~(uint32_t)(kSlotAlign - 1);
}

__device__ __forceinline__ unsigned long long seg_of_pad(const unsigned long long (&sg)[GCN10_N_RASTERS + 1], uint32_t r)
{
    unsigned long long v = 0;
#pragma unroll
    for (uint32_t q = 0; q <= (uint32_t)GCN10_N_RASTERS; q++)
        v = q == r ? sg[q] : v;
    return v;
}

__global__ __launch_bounds__(kPlaceThreads) void deflate_place_kernel(const TileJob job, uint32_t tiles_per_raster,
                                                                      uint32_t seg_align)
{
    __shared__ unsigned long long wave_tot[kPlaceThreads / 64];
    __shared__ unsigned long long P[GCN10_N_RASTERS + 1];       // unaligned prefix at each raster's first stream
    __shared__ unsigned long long seg[GCN10_N_RASTERS + 1];     // where each raster's extent starts
    constexpr uint32_t kMaxK = 40;                              // entries per thread kept in registers (40 960 streams)
    const uint32_t t = threadIdx.x;
    const uint32_t n = job.n_tiles;
    const uint32_t n_rasters = n / tiles_per_raster;
    const uint32_t K = (n + kPlaceThreads - 1) / kPlaceThreads;
    const uint32_t i0 = t * K < n ? t * K : n, i1 = i0 + K < n ? i0 + K : n;
    const uint2 *tab = reinterpret_cast<const uint2 *>(job.table);

    // one pass over the table: the thread's entries stay in registers (K <= kMaxK: strips of up to 4 096 rows of a
    // 36000-px block; larger launches re-read), its sum, and its sum up to the raster boundary it may contain
    // (K < tiles_per_raster whenever a raster has more tiles than a thread has entries: at most one boundary)
    uint32_t need[kMaxK];
    unsigned long long sum = 0, sum_at_boundary = 0;
    uint32_t boundary = 0xffffffffu;                            // the raster that starts inside this thread's range
    const bool in_regs = K <= kMaxK && K < tiles_per_raster;
    if (in_regs) {
#pragma unroll
        for (uint32_t k = 0; k < kMaxK; k++) {
            need[k] = 0;
            if (k < K && i0 + k < i1) {
                const uint2 e = tab[i0 + k];
                need[k] = place_need(e.x, e.y) | (e.x == kAliasSlot ? 0x80000000u : 0u);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kMaxK; k++)
            if (k < K && i0 + k < i1) {
                if ((i0 + k) % tiles_per_raster == 0) {
                    boundary = (i0 + k) / tiles_per_raster;
                    sum_at_boundary = sum;
                }
                sum += need[k] & 0x7fffffffu;
            }
    }
    else {
        for (uint32_t i = i0; i < i1; i++) {
            const uint2 e = tab[i];
            sum += place_need(e.x, e.y);
        }
    }
    // exclusive scan over the workgroup
    unsigned long long incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, 64);
        if ((int)(t & 63u) >= off)
            incl += up;
    }
    if ((t & 63u) == 63u)
        wave_tot[t >> 6] = incl;
    __syncthreads();
    unsigned long long base = incl - sum, total = 0;
    for (uint32_t w = 0; w < kPlaceThreads / 64; w++) {
        if (w < (t >> 6))
            base += wave_tot[w];
        total += wave_tot[w];
    }
    if (in_regs) {
        if (boundary != 0xffffffffu)
            P[boundary] = base + sum_at_boundary;
    }
    else {
        unsigned long long running = base;
        for (uint32_t i = i0; i < i1; i++) {
            const uint2 e = tab[i];
            if (i % tiles_per_raster == 0)
                P[i / tiles_per_raster] = running;
            running += place_need(e.x, e.y);
        }
    }
    if (t == 0)
        P[n_rasters] = total;
    __syncthreads();
    // every thread works the extents out for itself (18 steps on LDS values: cheaper than a third barrier)
    unsigned long long my_seg[GCN10_N_RASTERS + 1];
    {
        const unsigned long long a = seg_align;
        my_seg[0] = 0;
#pragma unroll
        for (uint32_t r = 1; r <= (uint32_t)GCN10_N_RASTERS; r++) {
            const unsigned long long end = r <= n_rasters ? my_seg[r - 1] + (P[r] - P[r - 1]) : 0ull;
            my_seg[r] = r < n_rasters ? (end + a - 1) / a * a : end;
        }
        if (t == 0) {
            unsigned long long used = 0;
#pragma unroll
            for (uint32_t r = 0; r <= (uint32_t)GCN10_N_RASTERS; r++) {
                seg[r] = my_seg[r];
                used = r == n_rasters ? my_seg[r] : used;
            }
            *job.cursor = used;
        }
    }
    {
        unsigned long long running = base;
        const uint32_t r0 = i0 < n ? i0 / tiles_per_raster : 0u;
        auto seg_of = [&](uint32_t r) -> unsigned long long {
            unsigned long long v = 0;
#pragma unroll
            for (uint32_t q = 0; q <= (uint32_t)GCN10_N_RASTERS; q++)
                v = q == r ? my_seg[q] : v;
            return v;
        };
        if (in_regs) {
            // (at most two rasters in a thread's range)
            const unsigned long long sa = seg_of(r0), pa = P[r0 < n_rasters ? r0 : 0], sb = seg_of(r0 + 1),
                                     pb = P[r0 + 1 <= n_rasters ? r0 + 1 : 0];
#pragma unroll
            for (uint32_t k = 0; k < kMaxK; k++)
                if (k < K && i0 + k < i1) {
                    const uint32_t i = i0 + k;
                    const bool second = i / tiles_per_raster != r0;
                    const uint32_t nd = need[k] & 0x7fffffffu;
                    if (!(need[k] & 0x80000000u)) {
                        const unsigned long long off = (second ? sb : sa) + (running - (second ? pb : pa));
                        const bool fits = off + nd <= job.arena_cap;
                        job.table[(size_t)i * 2] = fits ? (uint32_t)off : 0xffffffffu;
                        if (!fits)
                            job.table[(size_t)i * 2 + 1] = 0u;
                    }
                    running += nd;
                }
        }
        else {
            for (uint32_t i = i0; i < i1; i++) {
                const uint2 e = tab[i];
                const uint32_t r = i / tiles_per_raster;
                const uint32_t nd = place_need(e.x, e.y);
                if (e.x != kAliasSlot) {
                    const unsigned long long off = seg_of(r) + (running - P[r]);
                    const bool fits = off + nd <= job.arena_cap;
                    job.table[(size_t)i * 2] = fits ? (uint32_t)off : 0xffffffffu;
                    if (!fits)
                        job.table[(size_t)i * 2 + 1] = 0u;
                }
                running += nd;
            }
        }
    }
    // the pad between a raster's last stream and the next raster's extent reads as zeros, and so do the
    // bytes from the last stream's end to the next multiple of the alignment (an O_DIRECT write reads them):
    // wave w zeroes behind rasters w, w + 16
    for (uint32_t r = t >> 6; r < n_rasters; r += kPlaceThreads / 64) {
        const unsigned long long from = seg_of_pad(my_seg, r) + (P[r + 1] - P[r]);
        unsigned long long to = r + 1 < n_rasters ? seg_of_pad(my_seg, r + 1) : (from + seg_align - 1) / seg_align * seg_align;
        if (to > job.arena_cap)
            to = job.arena_cap;
        for (unsigned long long o = from + (unsigned long long)(t & 63u) * 16u; o + 16u <= to; o += 64u * 16u)
            *reinterpret_cast<gcn10::u32x4 *>(job.arena + o) = gcn10::u32x4{ 0u, 0u, 0u, 0u };
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
    const uint32_t slot = job.table[(size_t)blockIdx.x * 2];       // placed by deflate_place_kernel
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

}  // namespace

namespace gcn10 {

int deflate_workspace(gcn10_gpu_ctx *ctx, size_t need)
{
    if (need <= ctx->deflate_ws_cap)
        return GCN10_OK;
    HIP_TRY(hipDeviceSynchronize());            // the old workspace may still be in use
    if (ctx->deflate_ws)
        HIP_TRY(hipFree(ctx->deflate_ws));
    ctx->deflate_ws = nullptr;
    ctx->deflate_ws_cap = 0;
    HIP_TRY(hipMalloc(&ctx->deflate_ws, need));
    ctx->deflate_ws_cap = need;
    return GCN10_OK;
}

int deflate_launch_codes(gcn10_gpu_ctx *ctx, const TileJob &job, uint32_t nblocks, hipStream_t s, bool place)
{
    static_assert(sizeof(Work) * kBuildThreads <= 160 * 1024, "code construction slices must fit LDS");
    if (ctx->deflate_wave_codes) {
        const uint32_t per_raster = job.across * job.down;
        hipLaunchKernelGGL(deflate_codes_wave_kernel, dim3((per_raster + kWavesPerBlock - 1) / kWavesPerBlock, nblocks / per_raster),
                           dim3(64 * kWavesPerBlock), 0, s, job);
    }
    else {
        if (!ctx->codes_ready) {
            // more than 64 KiB of dynamic LDS has to be asked for, once per device
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_codes_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(Work) * kBuildThreads)));
            ctx->codes_ready = true;
        }
        hipLaunchKernelGGL(deflate_codes_kernel, dim3((nblocks + kBuildThreads - 1) / kBuildThreads),
                           dim3(kBuildThreads), sizeof(Work) * kBuildThreads, s, job);
    }
    if (place)
        hipLaunchKernelGGL(deflate_place_kernel, dim3(1), dim3(kPlaceThreads), 0, s, job, job.across * job.down,
                           (uint32_t)ctx->arena_segment_align);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

}  // namespace gcn10

extern "C" {

size_t gcn10_gpu_deflate_arena_bound(int W, int rows, int n_rasters)
{
    if (W <= 0 || rows <= 0 || n_rasters <= 0)
        return 0;
    const size_t across = ((size_t)W + kTile - 1) / kTile, down = ((size_t)rows + kTile - 1) / kTile;
    const size_t slot = ((size_t)kMaxStream + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
    // + the pads that bring every raster's extent to a multiple of the largest segment alignment
    return across * down * (size_t)n_rasters * slot + (size_t)n_rasters * 4096u;
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
    job.codes_stop = (uint32_t)ctx->codes_stop;
    job.rasters = rasters_dev;
    job.arena = arena_dev;
    job.table = table_dev;
    job.sizes = table_dev;          // pass B' places in the table itself
    job.chunk_tot = nullptr;
    job.n_chunks = 0;
    job.cursor = cursor_dev;
    job.W = (uint32_t)W;
    job.rows = (uint32_t)rows;
    job.across = ((uint32_t)W + kTile - 1) / kTile;
    job.down = ((uint32_t)rows + kTile - 1) / kTile;
    // stream offsets are 32-bit table entries with 0xfffffffe / 0xffffffff reserved
    if (arena_cap >= 0xfffffffeull)
        return fail(GCN10_E_INVAL, "tile encoder: an arena of %zu bytes does not fit 32-bit stream offsets "
                                   "(use fewer rows per strip)", arena_cap);
    job.arena_cap = arena_cap;
    const uint64_t nblocks = (uint64_t)job.across * job.down * (uint64_t)n_rasters;
    if (nblocks > 0x7fffffffull)
        return fail(GCN10_E_INVAL, "gcn10_gpu_deflate_strip: too many tiles");
    job.n_tiles = (uint32_t)nblocks;

    // per-tile statistics and code books: workspace owned by the context
    const size_t need = (size_t)nblocks * ((size_t)kHistWords * 4 + (size_t)kBookBytes);
    rc = gcn10::deflate_workspace(ctx, need);
    if (rc)
        return rc;
    job.hist = reinterpret_cast<uint32_t *>(ctx->deflate_ws);
    job.books = reinterpret_cast<uint8_t *>(ctx->deflate_ws) + (size_t)nblocks * kHistWords * 4;

    static_assert(sizeof(SharedC<false>) <= 160 * 1024, "tile + output image must fit the CU's 160 KiB of LDS");
    static_assert(sizeof(SharedC<true>) <= 80 * 1024, "two small-stream workgroups must fit one CU");
    static_assert(kBookBytes % 4 == 0, "code books are dword aligned");
    if (!ctx->deflate_ready) {
        // more than 64 KiB of dynamic LDS has to be asked for, once per device
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_stats_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedA)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_emit_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedC<true>)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(deflate_emit_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SharedC<false>)));
        ctx->deflate_ready = true;
    }
    hipStream_t s = as_stream(ctx, stream);
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(deflate_stats_kernel, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedA), s, job);
    rc = gcn10::deflate_launch_codes(ctx, job, (uint32_t)nblocks, s);
    if (rc)
        return rc;
    hipLaunchKernelGGL(deflate_emit_kernel<true>, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedC<true>), s,
                       job);
    hipLaunchKernelGGL(deflate_emit_kernel<false>, dim3((uint32_t)nblocks), dim3(kTile), sizeof(SharedC<false>),
                       s, job);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}
}  // extern "C"

// gcn10_inflate.hip -- zlib/DEFLATE decoding of landcover tiles on the GPU.
//
// What it replaces: GDALRasterIO inside load_raster() (/root/reference/src/raster.c:167-176)
// inflates the ESA WorldCover tiles on the host before the pixel loop sees a byte; the
// VRT's sources are 36000x36000 DEFLATE GeoTIFFs with 1024x1024 internal tiles
// (/root/reference/landcover/esa_worldcover_2021.vrt:266-270), 1.3 GB decoded per block.
// Here the *compressed* tiles cross PCIe (a few % of the raw bytes) and are decoded in
// HBM: a DEFLATE stream is serial, a block's ~1300 tiles are not, so one wavefront decodes
// one tile and a launch decodes them all.
//
// One workgroup of two wavefronts per stream (RFC 1950 wrapper, RFC 1951 stored / fixed /
// dynamic blocks): the DECODER wave turns bits into tokens (literals, match, stored run), the
// COPIER wave carries tokens out on the window; they swap halves of a 2 x 64-token ring at a
// barrier, so the bit decode never waits for a copy.  Inside a block the decoder's lanes decode
// speculatively -- lane k the token that would start at bit P + k -- and a scalar walk follows
// the real chain through them (decode_batch), which shrinks the serial part of the decode to a
// v_readlane and a few scalar instructions per token.
//   input    each decoder lane holds one dword of a 256-byte piece of the stream (plus the
//            next piece, already in flight); the bit reader takes dwords with v_readlane at a
//            wave-uniform index, so the decode state lives in scalar registers
//   tables   canonical codes are sorted by (length, symbol) with ballots; the 10-bit
//            (literal/length) and 9-bit (distance) lookup tables are filled entry-parallel,
//            16 resp. 8 entries per lane; longer codes take a bit-serial canonical walk
//   window   the last 32 KiB of output live in LDS (reads after writes are ordered there);
//            a match is copied by all copier lanes, 64 bytes per step, also when it overlaps
//            itself; every 16 KiB the finished half goes to HBM as 16 B per lane
//   output   each tile decodes into its own linear slot; untile_kernel then copies the
//            wanted window of every tile into the row-major landcover block.
// Malformed streams end with a status code, never with a wild access or an endless loop
// (every trip of every loop consumes input bits, and input is bounded).
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

// The LDS ring holds the last 16 KiB of output; DEFLATE distances reach 32 KiB, and the rare match that
// starts further back than the ring reaches is read from the tile's slot in HBM, where every byte
// older than kFlush + 258 already is (copy_match).  16 KiB instead of 32 lets six streams share a CU
// (a block's ~1300-1370 tiles then decode in one round of 1536 slots instead of 1024 + a ragged rest).
// Round 3: 8 KiB (GCN10_INFLATE_WINDOW; flush unit = half of it, sub-batches of the copier a quarter).  A
// workgroup then needs 15.6 KB of LDS instead of 23.6: five decoder workgroups per CU (a block's 1 296 streams)
// take 78 KB and leave room for a 77 KB tile workgroup of the encoder's pass F-A beside them -- with 16 KiB rings
// the decoder of block N+1 and the encoder of block N, which the round-3 pipeline runs side by side, took turns
// on every CU.  Measured in the pipeline, steady state, 72 blocks, three alternating runs on one box
// (profiles/r03/inflate_window_pipeline_72_blocks.txt): noisy blocks 0.0268-0.0282 -> 0.0215-0.0239 s with two
// workers per GPU (0.0304 -> 0.0265 with one), patchy ones unchanged (0.0099-0.0104).  The decoder alone is
// 0-9 % slower with the smaller ring (more matches read back from HBM: inflate_window_ab.txt, whose 16-block
// pipeline rows were start-up dominated and showed nothing).
// Invariant (copy_match): kFlush + kSubCap + 258 <= kWindow.
#ifndef GCN10_INFLATE_WINDOW
#define GCN10_INFLATE_WINDOW 8192
#endif
constexpr int kWindow = GCN10_INFLATE_WINDOW;
constexpr int kWindowMask = kWindow - 1;
constexpr int kFlush = kWindow / 2;
constexpr int kBatch = 64;              // token words per hand-over from the decoder to the copier
constexpr int kCand = 4;               // candidate start bits per lane: a window of 64 * kCand bits
constexpr int kLitRoot = 10;
constexpr int kDistRoot = 9;

enum {
    kOk = 0,
    kErrHeader = 1,         // not a zlib stream (CM != 8, FDICT set, bad check bits)
    kErrBlockType = 2,
    kErrStored = 3,         // LEN != ~NLEN
    kErrLengths = 4,        // bad code-length sequence or over-subscribed code
    kErrCode = 5,           // a bit pattern that is no code of the current block
    kErrDistance = 6,       // distance reaches before the start of the output
    kErrInput = 7,          // ran past the end of the compressed bytes
    kErrWindow = 8,         // the descriptor's window does not lie inside its chunk
};

// the wanted window must lie inside the decoded chunk (and the chunk inside its slot)
__device__ __forceinline__ bool window_ok(uint32_t out_len, uint32_t chunk_w, uint32_t src_x, uint32_t src_y,
                                          uint32_t copy_w, uint32_t copy_h, uint32_t slot_bytes)
{
    if (copy_w == 0 || copy_h == 0)
        return true;
    if (chunk_w == 0 || out_len > slot_bytes || src_x > chunk_w || copy_w > chunk_w - src_x)
        return false;
    const unsigned long long last = (unsigned long long)(src_y + copy_h - 1u) * chunk_w + src_x + copy_w;
    return (unsigned long long)src_y + copy_h <= 0xffffffffull && last <= out_len;
}

struct Code {               // canonical code, by length
    uint16_t count[16];
    uint16_t first[16];     // first code of each length
    uint16_t offs[16];      // index of its symbol in sorted[]
};

struct Shared {
    uint8_t window[kWindow];
    uint32_t lit_tab[1 << kLitRoot];        // decoded entries (below), 0 = longer code or none
    uint32_t dist_tab[1 << kDistRoot];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint8_t lens[320];
    Code lit, dist;
    // decoder wave -> copier wave: two batches of tokens (below), one being filled while the
    // other is carried out
    uint32_t ring[2][kBatch];
    uint32_t count[2];
    uint32_t stop;              // the trip after which both waves leave the loop
    uint32_t err;
};

struct Reader {
    const uint32_t *in;     // 16-byte aligned start of the stream
    uint32_t n_dwords;      // dwords that hold stream bytes
    uint32_t ip;            // next dword to take
    uint32_t chunk;         // index of the 64-dword piece held in `cur`
    uint32_t cur, nxt;      // per lane
    uint32_t prv;           // per lane: the piece before `cur` (the bit buffer may still hold bits of it)
    unsigned long long bb;  // bit buffer, LSB first
    uint32_t bc;            // valid bits in bb
    int lane;
};

__device__ __forceinline__ uint32_t load_piece(const Reader &r, uint32_t chunk)
{
    const uint32_t i = chunk * 64u + (uint32_t)r.lane;
    return i < r.n_dwords ? r.in[i] : 0u;
}

__device__ __forceinline__ void reader_seek(Reader &r, uint32_t dword)
{
    r.ip = dword;
    r.chunk = dword >> 6;
    r.cur = load_piece(r, r.chunk);
    r.nxt = load_piece(r, r.chunk + 1);
    r.prv = 0;
    r.bb = 0;
    r.bc = 0;
}

// The stream is held a piece (64 dwords, one per lane) at a time: the piece before the
// current one, the current one and the next (already in flight).  Pieces only ever advance.
__device__ __forceinline__ void ensure_piece(Reader &r, uint32_t c)
{
    if (c == r.chunk + 1u) {                    // wave-uniform
        r.prv = r.cur;
        r.cur = r.nxt;
        r.chunk = c;
        r.nxt = load_piece(r, c + 1);
    }
}

// dword q of the stream as a wave-uniform value (q lies in one of the three pieces)
__device__ __forceinline__ uint32_t stream_dword(const Reader &r, uint32_t q)
{
    const uint32_t in_prv = (uint32_t)__builtin_amdgcn_readlane((int)r.prv, (int)(q & 63u));
    const uint32_t in_cur = (uint32_t)__builtin_amdgcn_readlane((int)r.cur, (int)(q & 63u));
    const uint32_t in_nxt = (uint32_t)__builtin_amdgcn_readlane((int)r.nxt, (int)(q & 63u));
    const uint32_t c = q >> 6;
    return c == r.chunk ? in_cur : (c + 1u == r.chunk ? in_prv : in_nxt);
}

__device__ __forceinline__ uint32_t take_dword(Reader &r)
{
    ensure_piece(r, r.ip >> 6);
    const uint32_t v = stream_dword(r, r.ip);
    r.ip++;
    return v;
}

// at least 33 bits in the buffer afterwards
__device__ __forceinline__ void refill(Reader &r)
{
    if (r.bc <= 32) {
        r.bb |= (unsigned long long)take_dword(r) << r.bc;
        r.bc += 32;
    }
}

__device__ __forceinline__ uint32_t take_bits(Reader &r, uint32_t n)
{
    const uint32_t v = (uint32_t)r.bb & ((1u << n) - 1u);
    r.bb >>= n;
    r.bc -= n;
    return v;
}

// bytes of the stream consumed so far (whole bytes only)
__device__ __forceinline__ uint32_t reader_byte_pos(const Reader &r)
{
    return r.ip * 4u - r.bc / 8u;
}

__device__ __forceinline__ uint32_t uniform(uint32_t v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// Inclusive prefix sum over the 64 lanes with DPP adds.
__device__ __forceinline__ uint32_t wave_scan(uint32_t x)
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

// Sum over the 64 lanes, wave-uniform, with DPP adds (no LDS round trips): quads, rows of 16, then
// row_bcast:15 / row_bcast:31 carry the row totals to lane 63.
__device__ __forceinline__ uint32_t wave_total(uint32_t x)
{
    uint32_t r = x;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0xb1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x4e, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x141, 0xf, 0xf, false);     // row_half_mirror
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x140, 0xf, 0xf, false);     // row_mirror
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
}

// Sorts the n symbols of lens[] by (length, symbol) and derives first code / offset per
// length.  Returns false when the lengths over-subscribe the code space.
__device__ __forceinline__ bool sort_code(const uint8_t *lens, int n, uint16_t *sorted, Code &c, int lane)
{
    uint32_t cnt[16];
#pragma unroll
    for (int L = 0; L < 16; L++)
        cnt[L] = 0;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t mylen = s < n ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L < 16; L++)
            cnt[L] += (uint32_t)__builtin_popcountll(__ballot(mylen == (uint32_t)L));
    }
    uint32_t offs[16], first[16];
    int left = 1;
    bool ok = true;
    offs[0] = 0;
    first[0] = 0;
    offs[1] = 0;
    first[1] = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
        left = left * 2 - (int)cnt[L];
        if (left < 0)
            ok = false;
        if (L < 15) {
            offs[L + 1] = offs[L] + cnt[L];
            first[L + 1] = (first[L] + cnt[L]) << 1;
        }
    }
    if (lane < 16) {
        uint32_t cv = 0, fv = 0, ov = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            if (lane == L) {
                cv = cnt[L];
                fv = first[L];
                ov = offs[L];
            }
        }
        c.count[lane] = (uint16_t)cv;
        c.first[lane] = (uint16_t)fv;
        c.offs[lane] = (uint16_t)ov;
    }
    uint32_t run[16];
#pragma unroll
    for (int L = 0; L < 16; L++)
        run[L] = offs[L];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t mylen = s < n ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            const unsigned long long m = __ballot(mylen == (uint32_t)L);
            if (mylen == (uint32_t)L)
                sorted[run[L] + (uint32_t)__builtin_popcountll(m & below)] = (uint16_t)s;
            run[L] += (uint32_t)__builtin_popcountll(m);
        }
    }
    return ok;
}

// Table entries, indexed by the low ROOT bits of the bit buffer (codes arrive MSB first, so
// the buffer's bit 0 is a code's first bit).  Bits 0..3 of an entry = bits to consume, 0 = no
// code of at most ROOT bits starts here (longer code, or none).
//   literal/length table   bits 4..5 kind: 0 literals, 1 match length, 2 end of block
//     literals  bits 6..7 how many (1..3: as many whole literal codes as fit the ROOT bits),
//               bits 8..31 their bytes, first one lowest
//     length    bits 6..8 number of extra bits, bits 9..17 base length
//   distance table         bits 4..7 number of extra bits, bits 8..23 base distance
//   code-length table      bits 4..8 symbol
enum { kLiterals = 0, kLength = 1, kEndOfBlock = 2 };

__device__ __forceinline__ uint32_t length_entry(uint32_t sym, uint32_t bits)
{
    // sym = 257..285, RFC 1951 3.2.5
    const uint32_t s = sym - 257u;
    if (s > 28u)
        return 0;
    uint32_t base, eb = 0;
    if (s < 8u) {
        base = 3u + s;
    }
    else if (s == 28u) {
        base = 258u;
    }
    else {
        eb = (s - 4u) >> 2;
        base = 3u + ((4u + (s & 3u)) << eb);
    }
    return bits | (uint32_t)kLength << 4 | eb << 6 | base << 9;
}

__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t bits)
{
    if (sym > 29u)
        return 0;
    uint32_t base, eb = 0;
    if (sym < 4u) {
        base = 1u + sym;
    }
    else {
        eb = (sym - 2u) >> 1;
        base = 1u + ((2u + (sym & 1u)) << eb);
    }
    return bits | eb << 4 | base << 8;
}

// the code at the low bits of `v`, looking at no more than `avail` (<= ROOT) of them:
// symbol << 4 | length, or 0
template <int ROOT>
__device__ __forceinline__ uint32_t walk(uint32_t v, int avail, const uint32_t (&cnt)[ROOT + 1],
                                         const uint32_t (&first)[ROOT + 1], const uint32_t (&offs)[ROOT + 1],
                                         const uint16_t *sorted)
{
    uint32_t code = 0, e = 0;
#pragma unroll
    for (int L = 1; L <= ROOT; L++) {
        code = (code << 1) | ((v >> (L - 1)) & 1u);
        const uint32_t d = code - first[L];
        if (e == 0 && L <= avail && code >= first[L] && d < cnt[L])
            e = (uint32_t)sorted[offs[L] + d] << 4 | (uint32_t)L;
    }
    return e;
}

enum { kCodeLengthTable = 0, kLitLenTable = 1, kDistTable = 2 };

template <int ROOT, int WHICH>
__device__ __forceinline__ void fill_table(uint32_t *tab, const uint16_t *sorted, const Code &c, int lane)
{
    uint32_t cnt[ROOT + 1], first[ROOT + 1], offs[ROOT + 1];
#pragma unroll
    for (int L = 1; L <= ROOT; L++) {
        cnt[L] = c.count[L];
        first[L] = c.first[L];
        offs[L] = c.offs[L];
    }
    for (int idx = lane; idx < (1 << ROOT); idx += 64) {
        const uint32_t w = walk<ROOT>((uint32_t)idx, ROOT, cnt, first, offs, sorted);
        uint32_t e = 0;
        if (w != 0) {
            const uint32_t sym = w >> 4, bits = w & 15u;
            if (WHICH == kCodeLengthTable) {
                e = w;
            }
            else if (WHICH == kDistTable) {
                e = dist_entry(sym, bits);
            }
            else if (sym == 256u) {
                e = bits | (uint32_t)kEndOfBlock << 4;
            }
            else if (sym > 256u) {
                e = length_entry(sym, bits);
            }
            else {
                // up to three literals whose codes fit the ROOT bits together
                uint32_t bytes = sym, n = 1, used = bits;
#pragma unroll
                for (int more = 0; more < 2; more++) {
                    const uint32_t w2 = used < (uint32_t)ROOT && n == (uint32_t)more + 1u
                                            ? walk<ROOT>((uint32_t)idx >> used, ROOT - (int)used, cnt, first, offs, sorted)
                                            : 0u;
                    if (w2 != 0 && (w2 >> 4) < 256u) {
                        bytes |= (w2 >> 4) << (8 * n);
                        n++;
                        used += w2 & 15u;
                    }
                }
                e = used | (uint32_t)kLiterals << 4 | n << 6 | bytes << 8;
            }
        }
        tab[idx] = e;
    }
}

// a code longer than the table's root, or none: walk the canonical code bit by bit;
// symbol << 4 | length, or 0
__device__ __forceinline__ uint32_t slow_symbol(const Reader &r, const uint16_t *sorted, const Code &c)
{
    uint32_t code = 0;
    for (int L = 1; L < 16; L++) {
        code = (code << 1) | ((uint32_t)(r.bb >> (L - 1)) & 1u);
        const uint32_t f = c.first[L], n = c.count[L];
        if (code >= f && code - f < n)
            return (uint32_t)sorted[c.offs[L] + code - f] << 4 | (uint32_t)L;
    }
    return 0;
}

struct Output {
    uint8_t *out;           // this tile's slot in HBM, 16-byte aligned -- or, for a tile decoded in place, the first
                            // byte of its window in the destination raster
    uint32_t limit;         // bytes wanted
    uint32_t pos;           // bytes produced
    uint32_t flushed;       // bytes already in HBM (multiple of kFlush)
    // A tile decoded IN PLACE (round 3): the whole chunk is wanted, it is a power of two wide and its rows start on
    // 16-byte boundaries of the destination, so decoded byte p goes straight to row p / width, column p % width of the
    // raster -- no slot, no pass of untile_kernel over it (94 % of a block's tiles: all but its last row and column).
    uint32_t wshift;        // log2(chunk width); 0 = the linear slot
    uint32_t wmask;
    unsigned long long stride;
};

__device__ __forceinline__ uint8_t *out_at(const Output &o, uint32_t p)
{
    return o.wshift ? o.out + (size_t)(p >> o.wshift) * o.stride + (p & o.wmask) : o.out + p;
}

__device__ __forceinline__ void flush_half(Shared &sh, Output &o, int lane)
{
    const u32x4 *src = reinterpret_cast<const u32x4 *>(sh.window + (o.flushed & kWindowMask));
#pragma unroll 4
    for (int i = lane; i < kFlush / 16; i += 64)
        *reinterpret_cast<u32x4 *>(out_at(o, o.flushed + 16u * (uint32_t)i)) = src[i];      // (16 bytes never leave a row: widths are multiples of 16)
    o.flushed += kFlush;
}

// `ahead`: bytes of the ring from this token's first byte up to the furthest byte already written (the
// token's own length when tokens are carried out strictly in order; more when later tokens of the
// sub-batch have been carried out first): the ring bytes that far behind have been overwritten.
__device__ __forceinline__ void copy_match(Shared &sh, Output &o, uint32_t len, uint32_t dist, uint32_t ahead, int lane)
{
    const uint32_t from = o.pos - dist;
    if (dist + ahead > (uint32_t)kWindow) {
        // the source starts before what the ring still holds: all of it has been flushed to the slot
        // (from + len < flushed + kFlush + kSubCap - kWindow + len <= flushed: a sub-batch starts with fewer than
        // kFlush bytes pending and ends at most kSubCap further, and kFlush + kSubCap + 258 <= kWindow)
        // -- wait for those stores, read it back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint8_t v[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t k = (uint32_t)lane + 64u * (uint32_t)i;
            v[i] = k < len ? *out_at(o, from + k) : (uint8_t)0;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t k = (uint32_t)lane + 64u * (uint32_t)i;
            if (k < len)
                sh.window[(o.pos + k) & kWindowMask] = v[i];
        }
    }
    else if (len <= 64u && dist >= len) {
        // the usual case: one step, source and destination apart
        if ((uint32_t)lane < len)
            sh.window[(o.pos + (uint32_t)lane) & kWindowMask] = sh.window[(from + (uint32_t)lane) & kWindowMask];
    }
    else if (dist >= len) {
        // source and destination apart, up to 258 bytes: all the reads first, then all the writes
        // (one LDS round trip instead of five)
        uint8_t v[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t k = (uint32_t)lane + 64u * (uint32_t)i;
            v[i] = k < len ? sh.window[(from + k) & kWindowMask] : (uint8_t)0;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t k = (uint32_t)lane + 64u * (uint32_t)i;
            if (k < len)
                sh.window[(o.pos + k) & kWindowMask] = v[i];
        }
    }
    else if (dist >= 64u) {
        // 64 bytes per step; a later step may read what an earlier one wrote (LDS keeps order)
        for (uint32_t base = 0; base < len; base += 64u) {
            const uint32_t k = base + (uint32_t)lane;
            if (k < len)
                sh.window[(o.pos + k) & kWindowMask] = sh.window[(from + k) & kWindowMask];
        }
    }
    else if (dist == 1u && len >= 16u && (o.pos & (uint32_t)kWindowMask) + len <= (uint32_t)kWindow) {
        // a long run of one byte (every constant stretch of a landcover row is one), not across the ring's end: the
        // byte four times in a dword, one dword per lane and up to three bytes behind them.  (Runs of a few bytes,
        // which i.i.d. pixels are full of, stay with the general form below: it is faster for them.)
        const uint32_t d_off = o.pos & (uint32_t)kWindowMask;
        const uint32_t v = (uint32_t)sh.window[from & kWindowMask] * 0x01010101u;
        const uint32_t n4 = len >> 2, r = len & 3u;
        typedef uint32_t u32_u __attribute__((aligned(1)));
        if ((uint32_t)lane < n4)
            *reinterpret_cast<u32_u *>(sh.window + d_off + 4u * (uint32_t)lane) = v;
        if ((uint32_t)lane < r)
            sh.window[d_off + 4u * n4 + (uint32_t)lane] = (uint8_t)v;
    }
    else {
        // the copy repeats the last `dist` bytes: lane l always writes pattern byte l mod dist
        // when the step is a multiple of dist
        uint32_t step = dist;
        while (step * 2u <= 64u)
            step *= 2u;
        const uint32_t q = (uint32_t)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)dist));
        const uint8_t v = sh.window[(from + ((uint32_t)lane - q * dist)) & kWindowMask];
        for (uint32_t base = 0; base < len; base += step) {
            const uint32_t k = base + (uint32_t)lane;
            if ((uint32_t)lane < step && k < len)
                sh.window[(o.pos + k) & kWindowMask] = v;
        }
    }
    o.pos += len;
}

struct TileIn {             // = gcn10_inflate_tile
    unsigned long long in_off;
    uint32_t in_len, out_len;
    uint32_t chunk_w, src_x, src_y, copy_w, copy_h, flags;       // GCN10_TILE_*
    unsigned long long dst_off;
};
static_assert(sizeof(TileIn) == sizeof(gcn10_inflate_tile), "TileIn mirrors the ABI struct");

// Tokens the decoder wave hands to the copier wave, one word each (a stored run takes two):
//   literals  how many (1..3) << 24 | their bytes, first one lowest
//   match     0x80000000 | (length - 3) << 16 | (distance - 1)
//   stored    0x40000000 | length (<= kWindow / 4), then a word with the offset of the bytes in the stream
constexpr uint32_t kTokMatch = 0x80000000u, kTokStored = 0x40000000u;

enum { kNeedHeader = 0, kInSymbols = 1, kInStored = 2, kDone = 3 };

struct Decoder {            // the decoder wave's state between batches (all wave-uniform)
    Reader r;
    uint32_t state;
    uint32_t P;             // kInSymbols: bit position in the stream (the reader's own bit state is stale)
    uint32_t pos, limit;    // bytes the tokens so far produce; bytes wanted
    uint32_t stored_left, stored_at;
    uint32_t err;
    bool last;              // the current block is the final one
    uint32_t in_len;
};

// Parses a block header (and builds its tables).  Leaves d.state = kInSymbols / kInStored, or
// kDone with d.err set.
__device__ __forceinline__ void begin_block(Shared &sh, Decoder &d, int lane)
{
    Reader &r = d.r;
    refill(r);
    d.last = take_bits(r, 1) != 0;
    const uint32_t type = take_bits(r, 2);
    if (type == 0) {
        // stored: skip to the byte boundary, LEN, NLEN, LEN bytes
        take_bits(r, r.bc & 7u);
        refill(r);
        const uint32_t len = take_bits(r, 16);
        refill(r);
        const uint32_t nlen = take_bits(r, 16);
        if ((len ^ nlen) != 0xffffu) {
            d.err = kErrStored;
            d.state = kDone;
            return;
        }
        d.stored_at = reader_byte_pos(r);
        d.stored_left = len;
        if (d.stored_at + len > d.in_len) {
            d.err = kErrInput;
            d.state = kDone;
            return;
        }
        d.state = kInStored;
        return;
    }
    if (type == 3) {
        d.err = kErrBlockType;
        d.state = kDone;
        return;
    }
    int n_lit, n_dist;
    if (type == 1) {
        // fixed code: lengths 8 / 9 / 7 / 8, 30 distance codes of 5 bits (RFC 1951 3.2.6)
        for (int i = lane; i < 288; i += 64)
            sh.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
        if (lane < 32)
            sh.lens[288 + lane] = 5;
        n_lit = 288;
        n_dist = 32;
    }
    else {
        n_lit = (int)take_bits(r, 5) + 257;
        n_dist = (int)take_bits(r, 5) + 1;
        const int n_cl = (int)take_bits(r, 4) + 4;
        if (n_lit > 286 || n_dist > 30) {
            d.err = kErrLengths;
            d.state = kDone;
            return;
        }
        // code-length code: 19 symbols of up to 7 bits, table over 7 bits in lit_tab
        if (lane < 19)
            sh.lens[lane] = 0;
        for (int i = 0; i < n_cl; i++) {
            refill(r);
            const uint32_t v = take_bits(r, 3);
            // i: 0 1 2 3 | 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 -> 16 17 18 0 | 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            const int sym = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? (19 - i) / 2 : 8 + (i - 4) / 2;
            if (lane == 0)
                sh.lens[sym] = (uint8_t)v;
        }
        if (!sort_code(sh.lens, 19, sh.lit_sorted, sh.lit, lane)) {
            d.err = kErrLengths;
            d.state = kDone;
            return;
        }
        fill_table<7, kCodeLengthTable>(sh.lit_tab, sh.lit_sorted, sh.lit, lane);
        // the n_lit + n_dist code lengths, run-length coded
        int have = 0;
        uint32_t prev = 0;
        const int total = n_lit + n_dist;
        while (have < total) {
            refill(r);
            const uint32_t e = uniform(sh.lit_tab[(uint32_t)r.bb & 127u]);
            if (e == 0) {
                d.err = kErrLengths;
                break;
            }
            take_bits(r, e & 15u);
            const uint32_t sym = e >> 4;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (have == 0) {
                    d.err = kErrLengths;
                    break;
                }
                rep = 3 + take_bits(r, 2);
                val = prev;
            }
            else if (sym == 17) {
                rep = 3 + take_bits(r, 3);
                val = 0;
            }
            else if (sym == 18) {
                rep = 11 + take_bits(r, 7);
                val = 0;
            }
            if (have + (int)rep > total) {
                d.err = kErrLengths;
                break;
            }
            for (uint32_t k = lane; k < rep; k += 64)
                sh.lens[have + k] = (uint8_t)val;       // lens[] is re-used: the 19 are in the table now
            have += (int)rep;
            prev = val;
        }
        if (d.err) {
            d.state = kDone;
            return;
        }
        // distance lengths follow the literal/length ones: move them to lens[288..]
        {
            const uint8_t dl = lane < n_dist ? sh.lens[n_lit + lane] : (uint8_t)0;
            if (lane < 32)
                sh.lens[288 + lane] = dl;
            for (int i = n_lit + lane; i < 288; i += 64)
                sh.lens[i] = 0;
        }
        n_lit = 288;
        n_dist = 32;
    }
    if (!sort_code(sh.lens, n_lit, sh.lit_sorted, sh.lit, lane) ||
        !sort_code(sh.lens + 288, n_dist, sh.dist_sorted, sh.dist, lane)) {
        d.err = kErrLengths;
        d.state = kDone;
        return;
    }
    fill_table<kLitRoot, kLitLenTable>(sh.lit_tab, sh.lit_sorted, sh.lit, lane);
    fill_table<kDistRoot, kDistTable>(sh.dist_tab, sh.dist_sorted, sh.dist, lane);
    d.P = r.ip * 32u - r.bc;
    d.state = kInSymbols;
}

// the scalar reader at bit P of the stream (P lies in the current piece or the one before)
__device__ __forceinline__ void seek_bits(Reader &r, uint32_t P)
{
    r.ip = P >> 5;
    r.bb = 0;
    r.bc = 0;
    refill(r);
    take_bits(r, P & 31u);
}

// One token by the scalar bit reader (codes longer than the tables' roots, or none).  The reader
// must stand at the token.  Returns false when the block or the stream ends here.
__device__ __forceinline__ bool scalar_token(Shared &sh, Decoder &d, uint32_t *ring, uint32_t &n, int lane)
{
    Reader &r = d.r;
    refill(r);
    uint32_t e = uniform(sh.lit_tab[(uint32_t)r.bb & ((1u << kLitRoot) - 1u)]);
    if (e == 0) {
        const uint32_t w = uniform(slow_symbol(r, sh.lit_sorted, sh.lit));
        const uint32_t sym = w >> 4;
        e = w == 0        ? 0u
            : sym < 256u  ? ((w & 15u) | 1u << 6 | sym << 8)
            : sym == 256u ? ((w & 15u) | (uint32_t)kEndOfBlock << 4)
                          : length_entry(sym, w & 15u);
        if (e == 0) {
            d.err = kErrCode;
            d.state = kDone;
            return false;
        }
    }
    take_bits(r, e & 15u);
    const uint32_t kind = (e >> 4) & 3u;
    uint32_t tok, outl;
    if (kind == (uint32_t)kLiterals) {
        outl = (e >> 6) & 3u;
        tok = outl << 24 | (e >> 8);
    }
    else if (kind == (uint32_t)kLength) {
        const uint32_t len = ((e >> 9) & 511u) + take_bits(r, (e >> 6) & 7u);
        refill(r);
        uint32_t de = uniform(sh.dist_tab[(uint32_t)r.bb & ((1u << kDistRoot) - 1u)]);
        if (de == 0) {
            const uint32_t w = uniform(slow_symbol(r, sh.dist_sorted, sh.dist));
            de = w == 0 ? 0u : dist_entry(w >> 4, w & 15u);
            if (de == 0) {
                d.err = kErrCode;
                d.state = kDone;
                return false;
            }
        }
        take_bits(r, de & 15u);
        const uint32_t dist = ((de >> 8) & 0xffffu) + take_bits(r, (de >> 4) & 15u);
        outl = len;
        tok = kTokMatch | (len - 3u) << 16 | (dist - 1u);
    }
    else {
        d.state = kNeedHeader;                      // end of block
        return false;
    }
    if (lane == 0)
        ring[n] = tok;
    n++;
    d.pos += outl;
    if (d.pos >= d.limit) {
        d.state = kDone;
        return false;
    }
    return true;
}

// The decoder wave: up to kBatch token words into `ring`; returns how many.  Every token
// produces output and output is bounded, every header consumes input and input is bounded, so
// the stream ends (d.state = kDone) whatever its bits are.
//
// Inside a block the 64 lanes decode SPECULATIVELY: lane k decodes the tokens that would start at
// bits P + k, P + 64 + k, ... of a 256-bit window (table lookups as gathers, all in vector
// registers; 8 bits of result per candidate, packed four to a register), and a scalar walk then follows
// the real chain -- start at lane 0, jump by each token's bit count -- collecting the tokens it
// passes.  The serial part of the decode is that walk: one v_readlane and a few scalar
// instructions per token instead of two dependent table lookups.
__device__ __forceinline__ uint32_t decode_batch(Shared &sh, Decoder &d, uint32_t *ring, int lane)
{
    Reader &r = d.r;
    uint32_t n = 0;
    while (n < (uint32_t)kBatch && d.state != (uint32_t)kDone) {
        if (d.state == (uint32_t)kNeedHeader) {
            if (d.last || d.pos >= d.limit) {
                d.state = kDone;
                break;
            }
            begin_block(sh, d, lane);
            continue;
        }
        if (d.state == (uint32_t)kInStored) {
            if (n + 2u > (uint32_t)kBatch)
                break;
            uint32_t len = d.stored_left < (uint32_t)(kWindow / 4) ? d.stored_left : (uint32_t)(kWindow / 4);    // = kSubCap
            if (len > d.limit - d.pos)
                len = d.limit - d.pos;
            if (len > 0) {
                if (lane == 0) {
                    ring[n] = kTokStored | len;
                    ring[n + 1] = d.stored_at;
                }
                n += 2;
                d.pos += len;
            }
            d.stored_at += len;
            d.stored_left -= len;
            if (d.stored_left == 0 || d.pos >= d.limit) {
                const uint32_t at = d.stored_at + d.stored_left;
                reader_seek(r, at >> 2);
                refill(r);
                take_bits(r, (at & 3u) * 8u);
                d.state = d.pos >= d.limit ? (uint32_t)kDone : (uint32_t)kNeedHeader;
            }
            continue;
        }
        // ---- kInSymbols: one window of 256 candidate start bits at d.P, four per lane ----
        const uint32_t q0 = d.P >> 5;
        ensure_piece(r, q0 >> 6);
        uint32_t D[11];
        if ((q0 & 63u) <= 53u) {
            // the usual case: all eleven dwords in one piece, one select for all of them
            const uint32_t piece = (q0 >> 6) == r.chunk ? r.cur : r.prv;
            const int l0 = (int)(q0 & 63u);
#pragma unroll
            for (int i = 0; i < 11; i++)
                D[i] = (uint32_t)__builtin_amdgcn_readlane((int)piece, l0 + i);
        }
        else {
#pragma unroll
            for (int i = 0; i < 11; i++)
                D[i] = stream_dword(r, q0 + (uint32_t)i);
        }
        const uint32_t rel = (d.P & 31u) + (uint32_t)lane;          // < 95
        const uint32_t o = rel >> 5, sft = rel & 31u;
        uint32_t tok[kCand];
        uint32_t packed = 0;        // per candidate 8 bits: bits it takes (< 64) | stop reason << 6
#pragma unroll
        for (int h = 0; h < kCand; h++) {
            // candidate at bit lane + 64 * h: dwords o + 2 * h .. + 2 of the eleven
            const uint32_t a = o == 0 ? D[2 * h] : o == 1 ? D[2 * h + 1] : D[2 * h + 2];
            const uint32_t b = o == 0 ? D[2 * h + 1] : o == 1 ? D[2 * h + 2] : D[2 * h + 3];
            const uint32_t c = o == 0 ? D[2 * h + 2] : o == 1 ? D[2 * h + 3] : D[2 * h + 4];
            const uint32_t lo = __builtin_amdgcn_alignbit(b, a, sft);   // bits of the candidate .. +31
            const uint32_t hi = __builtin_amdgcn_alignbit(c, b, sft);   //                     +32 .. +63
            const uint32_t e1 = sh.lit_tab[lo & ((1u << kLitRoot) - 1u)];
            const uint32_t cl = e1 & 15u, kind = (e1 >> 4) & 3u;
            uint32_t bits = cl;
            uint32_t tk = ((e1 >> 6) & 3u) << 24 | (e1 >> 8);
            uint32_t special = e1 == 0 ? 2u : (kind == (uint32_t)kEndOfBlock ? 1u : 0u);
            {
                // as if it were a match (harmless where it is not: the lookups stay in the tables)
                const uint32_t eb = (e1 >> 6) & 7u;
                const uint32_t len = ((e1 >> 9) & 511u) + ((lo >> cl) & ((1u << eb) - 1u));
                const uint32_t t = cl + eb;                             // <= 20
                const uint32_t x2 = __builtin_amdgcn_alignbit(hi, lo, t);
                const uint32_t de = sh.dist_tab[x2 & ((1u << kDistRoot) - 1u)];
                const uint32_t dl = de & 15u, deb = (de >> 4) & 15u;
                const uint32_t dist = ((de >> 8) & 0xffffu) + ((x2 >> dl) & ((1u << deb) - 1u));
                if (kind == (uint32_t)kLength && e1 != 0) {
                    bits = t + dl + deb;                                // <= 48
                    tk = kTokMatch | ((len - 3u) & 255u) << 16 | ((dist - 1u) & 0xffffu);
                    if (de == 0)
                        special = 2u;
                }
            }
            // a token the walk passes: its bits; one it stops at: an end of block (with its bits,
            // which the walk takes) or a code the tables do not hold (no bits: the scalar reader
            // starts at it)
            const uint32_t inf8 = special == 2u ? 0x80u : special == 1u ? (cl | 0x40u) : bits;
            packed |= inf8 << (8 * h);
            tok[h] = tk;
        }
        // ---- the chain: from bit 0 of the window, token by token.  One exit, no branches in the
        // body, one v_readlane per token: the loop is the serial core of the decoder.  The bytes the
        // tokens produce are summed afterwards (they are in the token words) ----
        // Written out, 17 instructions per token (the compiler's form of the same loop takes 24): the
        // start bit of the it-th token goes to lane `it` with one v_writelane (lane select in M0: two
        // different SGPR operands would break the constant-bus rule), and the three reasons to stop -- a
        // stop token, the end of the window, a full batch -- are chained through two scalar selects.
        const uint32_t room = (uint32_t)kBatch - n;        // 1..64: also keeps `it` within the 64 lanes
        uint32_t k = 0, it = 0, inf, t0, t1, m0_saved;
        uint32_t src = 0;
        asm volatile("s_mov_b32 %[m0s], m0\n"
                     "1:\n\t"
                     "v_readlane_b32 %[t0], %[packed], %[k]\n\t"       // lane k mod 64
                     "s_lshr_b32 %[t1], %[k], 3\n\t"
                     "s_and_b32 %[t1], %[t1], 24\n\t"
                     "s_mov_b32 m0, %[it]\n\t"
                     "s_lshr_b32 %[t0], %[t0], %[t1]\n\t"
                     "s_and_b32 %[inf], %[t0], 0xff\n\t"
                     "v_writelane_b32 %[src], %[k], m0\n\t"            // src[lane it] = k
                     "s_add_u32 %[it], %[it], 1\n\t"
                     "s_and_b32 %[t0], %[inf], 63\n\t"
                     "s_add_u32 %[k], %[k], %[t0]\n\t"
                     "s_cmp_lt_u32 %[inf], 64\n\t"
                     "s_cselect_b32 %[t0], %[k], 0x100\n\t"            // a stop token ends the walk
                     "s_cmp_lt_u32 %[t0], 0x100\n\t"
                     "s_cselect_b32 %[t0], %[it], %[room]\n\t"         // so does the end of the window
                     "s_cmp_lt_u32 %[t0], %[room]\n\t"                 // ... and a full batch
                     "s_cbranch_scc1 1b\n\t"
                     "s_mov_b32 m0, %[m0s]"                              // (M0 is the compiler's: put it back)
                     : [k] "+s"(k), [it] "+s"(it), [inf] "=&s"(inf), [t0] "=&s"(t0), [t1] "=&s"(t1), [src] "+v"(src),
                       [m0s] "=&s"(m0_saved)
                     : [packed] "v"(packed), [room] "s"(room)
                     : "scc");
        static_assert(kCand == 4, "the walk's window end is written as 0x100");
        uint32_t stop = inf >> 6;
        const uint32_t last_bits = inf & 63u;
        uint32_t n_new = stop != 0u ? it - 1u : it;        // a stop token is not a token of the stream
        {
            const int sel = (int)((src & 63u) << 2);
            uint32_t mine = 0;
#pragma unroll
            for (int h = 0; h < kCand; h++) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)tok[h]);
                mine = (src >> 6) == (uint32_t)h ? m : mine;
            }
            const bool kept = (uint32_t)lane < n_new;
            if (kept)
                ring[n + (uint32_t)lane] = mine;
            // bytes of the kept tokens: a match says its length, literals their count
            const uint32_t outl = !kept ? 0u : (mine & kTokMatch) ? ((mine >> 16) & 255u) + 3u : (mine >> 24);
            const uint32_t total = wave_total(outl);
            if (d.pos + total >= d.limit) {
                // the tile ends inside this window (once per stream): keep the tokens up to the one
                // that reaches the limit and stand behind it, exactly -- nothing behind it counts, not
                // even the end-of-block code the walk may have taken
                const uint32_t cum = wave_scan(outl);
                const uint32_t before = (uint32_t)__builtin_popcountll(__ballot(kept && d.pos + cum < d.limit));
                const uint32_t keep = before + 1u < n_new ? before + 1u : n_new;
                const uint32_t k_next = (uint32_t)__builtin_amdgcn_readlane((int)src, (int)(keep & 63u));
                k = keep < n_new ? k_next : (stop == 1u ? k - last_bits : k);
                d.pos += (uint32_t)__builtin_amdgcn_readlane((int)cum, (int)((keep - 1u) & 63u));
                n_new = keep;
                stop = 4u;
            }
            else {
                d.pos += total;
            }
            n += n_new;
        }
        d.P += k;
        if (stop == 4u) {
            d.state = kDone;
        }
        else if (stop == 1u) {
            // end of block (its code is taken): the next header is read by the scalar reader
            seek_bits(r, d.P);
            d.state = kNeedHeader;
        }
        else if (stop == 2u) {
            // a code the tables do not hold (or none): that one token by the scalar reader
            if (n < (uint32_t)kBatch) {
                seek_bits(r, d.P);
                if (scalar_token(sh, d, ring, n, lane) || d.state == (uint32_t)kNeedHeader)
                    d.P = r.ip * 32u - r.bc;
            }
            else {
                break;                              // no room: the next batch starts with it
            }
        }
        if (d.state == (uint32_t)kDone)
            seek_bits(r, d.P);                      // (for the end-of-input check)
    }
    return n;
}

// A match of 65..258 bytes whose source and destination ranges lie apart in the ring and do not cross its end
// (offsets s_off, d_off): one dword per lane (unaligned LDS access is enabled on this platform), one byte per lane
// for the last 0..3 bytes, ONE wait for those two reads, two writes.  copy_match()'s form of the same copy is five
// predicated byte reads and five predicated byte writes, ~100 instructions (cycle counters, round 3: 900 cycles per
// 258-byte match, and a patchy tile's decode waits for its copier).
__device__ __forceinline__ void copy_long(Shared &sh, uint32_t s_off, uint32_t d_off, uint32_t len, int lane)
{
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)sh.window;
    const uint32_t n4 = len >> 2, r = len & 3u;                 // whole dwords (16..64), bytes behind them
    const unsigned long long m4 = n4 >= 64u ? ~0ull : (1ull << n4) - 1ull, mr = (1ull << r) - 1ull;
    const uint32_t a4 = base + s_off + 4u * (uint32_t)lane, b4 = base + d_off + 4u * (uint32_t)lane;
    const uint32_t a1 = base + s_off + 4u * n4 + (uint32_t)lane, b1 = base + d_off + 4u * n4 + (uint32_t)lane;
    uint32_t v0, v1;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m4]\n\t"
                 "ds_read_b32 %[v0], %[a4]\n\t"
                 "s_mov_b64 exec, %[mr]\n\t"
                 "ds_read_u8 %[v1], %[a1]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_write_b8 %[b1], %[v1]\n\t"
                 "s_mov_b64 exec, %[m4]\n\t"
                 "ds_write_b32 %[b4], %[v0]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [v0] "=&v"(v0), [v1] "=&v"(v1), [sv] "=&s"(sv)
                 : [a4] "v"(a4), [b4] "v"(b4), [a1] "v"(a1), [b1] "v"(b1), [m4] "s"(m4), [mr] "s"(mr)
                 : "memory");
}

// A batch of tokens carried out one by one, in order (batches that hold a stored run).
// Returns 0, or the reason a token cannot be carried out (a distance before the start).
__device__ __forceinline__ uint32_t copy_batch_serial(Shared &sh, Output &o, const uint32_t *ring, uint32_t n,
                                               const uint8_t *stream, int lane)
{
    const uint32_t tk = (uint32_t)lane < n ? ring[lane] : 0u;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)tk, (int)i);
        if (o.pos >= o.limit)
            break;
        if (a & kTokMatch) {
            uint32_t len = ((a >> 16) & 255u) + 3u;
            const uint32_t dist = (a & 0xffffu) + 1u;
            if (dist > o.pos)
                return kErrDistance;
            if (len > o.limit - o.pos)
                len = o.limit - o.pos;
            copy_match(sh, o, len, dist, len, lane);
        }
        else if (a & kTokStored) {
            const uint32_t len = a & 0xffffu;
            i++;
            const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)tk, (int)i);
            for (uint32_t k = lane; k < len; k += 64)
                sh.window[(o.pos + k) & kWindowMask] = stream[at + k];
            o.pos += len;
        }
        else {
            const uint32_t cnt = a >> 24;
            if ((uint32_t)lane < cnt)
                sh.window[(o.pos + (uint32_t)lane) & kWindowMask] = (uint8_t)(a >> (8 * lane));
            o.pos += cnt;
        }
        if (o.pos - o.flushed >= (uint32_t)kFlush)
            flush_half(sh, o, lane);
    }
    return 0;
}

// The copier wave: carries out a batch of tokens on the window and sends finished halves to HBM.
// Returns 0, or the reason a token cannot be carried out (a distance before the start).
//
// One token per lane.  A prefix sum of the tokens' lengths gives every token its place; the batch is
// taken in sub-batches of at most kSubCap bytes (the ring must keep what is not flushed yet), and in
// each sub-batch
//   A  all literal tokens write their 1..3 bytes at once,
//   B  all short matches whose source lies wholly before the sub-batch (nothing in it can change
//      their source) copy their bytes at once, one lane per token,
//   C  the remaining matches are carried out one by one, in order, by all 64 lanes (copy_match).
// A token carried out early only writes bytes of its own place, so the order of A, B and C among
// tokens that do not read each other's output is free; C runs in stream order, and by then every
// byte an earlier token produces is there.
constexpr uint32_t kSubCap = kWindow / 4;     // (declared near kWindow: kSubCapBytes)
constexpr uint32_t kShortMatch = 64;    // stage B takes matches up to this length (sixteen dword steps + up to three bytes)

__device__ __forceinline__ uint32_t copy_batch(Shared &sh, Output &o, const uint32_t *ring, uint32_t n,
                                               const uint8_t *stream, int lane, uint32_t diag = 0)
{
    const uint32_t tk = (uint32_t)lane < n ? ring[lane] : 0u;
    const bool valid = (uint32_t)lane < n;
    if (__ballot(valid && !(tk & kTokMatch) && (tk & kTokStored)) != 0ull)
        return copy_batch_serial(sh, o, ring, n, stream, lane);
    const bool is_match = (tk & kTokMatch) != 0u;
    const uint32_t dist = (tk & 0xffffu) + 1u;
    uint32_t done = 0;
    while (done < n && o.pos < o.limit) {
        const bool in = valid && (uint32_t)lane >= done;
        const uint32_t len = !in ? 0u : is_match ? ((tk >> 16) & 255u) + 3u : (tk >> 24);
        const uint32_t incl = wave_scan(len);
        const uint32_t off = incl - len;
        const uint32_t m = (uint32_t)__builtin_popcountll(__ballot(in && incl <= kSubCap));    // >= 1
        const bool mine = in && (uint32_t)lane < done + m;
        const uint32_t S = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)((done + m - 1u) & 63u));
        const uint32_t dst = o.pos + off;
        // bytes wanted of this token: none behind the limit, the one that reaches it is cut
        uint32_t l = !mine || dst >= o.limit ? 0u : (len < o.limit - dst ? len : o.limit - dst);
        if (__ballot(is_match && l > 0u && dist > dst) != 0ull)
            return kErrDistance;
        const uint32_t ahead = S - off;                     // ring bytes written from dst on, once the sub-batch is out
        const bool near = dist + ahead <= (uint32_t)kWindow;
        // (stage B copies with unaligned dword accesses that must stay inside the ring: neither range across its end)
        const uint32_t s_ring = (dst - dist) & (uint32_t)kWindowMask, d_ring = dst & (uint32_t)kWindowMask;
        const bool early = is_match && l > 0u && l <= kShortMatch && near && dist >= off + l &&
                           s_ring + l <= (uint32_t)kWindow && d_ring + l <= (uint32_t)kWindow;
        // ... and the longer ones of the same kind (a patchy tile is made of them: 258 bytes from the row above),
        // one after the other but with every lane at work: stage B2
        const bool early_long = is_match && l > kShortMatch && near && dist >= off + l &&
                                s_ring + l <= (uint32_t)kWindow && d_ring + l <= (uint32_t)kWindow;
        // A: literals
        if (!is_match && l > 0u) {
#pragma unroll
            for (uint32_t b = 0; b < 3u; b++)
                if (b < l)
                    sh.window[(dst + b) & kWindowMask] = (uint8_t)(tk >> (8u * b));
        }
        // B: matches of up to kShortMatch bytes that read nothing of this sub-batch, all at once, one lane per match:
        // up to eight dword steps (unaligned LDS access is enabled on this platform) and up to three single bytes, all
        // the reads, ONE wait, all the writes.  Written out: the masks are scalar (ballots), the steps differ only in
        // their immediate offsets, and the compiler would wait after every read.  Round 3: this stage took matches
        // of at most 8 bytes, byte by byte; carried out one by one in stage C a match costs ~35 instructions of a
        // wave that shares its SIMD's issue slots with the decoders (profiles/r03/decoder_cycle_counters.txt).
        if (__ballot(early) != 0ull) {
            const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)sh.window;
            const uint32_t n4 = early ? l >> 2 : 0u, r = early ? l & 3u : 0u;
            unsigned long long M[16], T[3];
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++)
                M[k] = __ballot(k < n4);
#pragma unroll
            for (uint32_t k = 0; k < 3u; k++)
                T[k] = __ballot(k < r);
            const uint32_t a = base + s_ring, b = base + d_ring, at = a + 4u * n4, bt = b + 4u * n4;
            uint32_t v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, t0, t1, t2;
            unsigned long long sv;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b64 exec, %[M0]\n\t"
                         "ds_read_b32 %[v0], %[a]\n\t"
                         "s_mov_b64 exec, %[M1]\n\t"
                         "ds_read_b32 %[v1], %[a] offset:4\n\t"
                         "s_mov_b64 exec, %[M2]\n\t"
                         "ds_read_b32 %[v2], %[a] offset:8\n\t"
                         "s_mov_b64 exec, %[M3]\n\t"
                         "ds_read_b32 %[v3], %[a] offset:12\n\t"
                         "s_mov_b64 exec, %[M4]\n\t"
                         "ds_read_b32 %[v4], %[a] offset:16\n\t"
                         "s_mov_b64 exec, %[M5]\n\t"
                         "ds_read_b32 %[v5], %[a] offset:20\n\t"
                         "s_mov_b64 exec, %[M6]\n\t"
                         "ds_read_b32 %[v6], %[a] offset:24\n\t"
                         "s_mov_b64 exec, %[M7]\n\t"
                         "ds_read_b32 %[v7], %[a] offset:28\n\t"
                         "s_mov_b64 exec, %[M8]\n\t"
                         "ds_read_b32 %[v8], %[a] offset:32\n\t"
                         "s_mov_b64 exec, %[M9]\n\t"
                         "ds_read_b32 %[v9], %[a] offset:36\n\t"
                         "s_mov_b64 exec, %[M10]\n\t"
                         "ds_read_b32 %[v10], %[a] offset:40\n\t"
                         "s_mov_b64 exec, %[M11]\n\t"
                         "ds_read_b32 %[v11], %[a] offset:44\n\t"
                         "s_mov_b64 exec, %[M12]\n\t"
                         "ds_read_b32 %[v12], %[a] offset:48\n\t"
                         "s_mov_b64 exec, %[M13]\n\t"
                         "ds_read_b32 %[v13], %[a] offset:52\n\t"
                         "s_mov_b64 exec, %[M14]\n\t"
                         "ds_read_b32 %[v14], %[a] offset:56\n\t"
                         "s_mov_b64 exec, %[M15]\n\t"
                         "ds_read_b32 %[v15], %[a] offset:60\n\t"
                         "s_mov_b64 exec, %[T0]\n\t"
                         "ds_read_u8 %[t0], %[at]\n\t"
                         "s_mov_b64 exec, %[T1]\n\t"
                         "ds_read_u8 %[t1], %[at] offset:1\n\t"
                         "s_mov_b64 exec, %[T2]\n\t"
                         "ds_read_u8 %[t2], %[at] offset:2\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_write_b8 %[bt], %[t2] offset:2\n\t"
                         "s_mov_b64 exec, %[T1]\n\t"
                         "ds_write_b8 %[bt], %[t1] offset:1\n\t"
                         "s_mov_b64 exec, %[T0]\n\t"
                         "ds_write_b8 %[bt], %[t0]\n\t"
                         "s_mov_b64 exec, %[M15]\n\t"
                         "ds_write_b32 %[b], %[v15] offset:60\n\t"
                         "s_mov_b64 exec, %[M14]\n\t"
                         "ds_write_b32 %[b], %[v14] offset:56\n\t"
                         "s_mov_b64 exec, %[M13]\n\t"
                         "ds_write_b32 %[b], %[v13] offset:52\n\t"
                         "s_mov_b64 exec, %[M12]\n\t"
                         "ds_write_b32 %[b], %[v12] offset:48\n\t"
                         "s_mov_b64 exec, %[M11]\n\t"
                         "ds_write_b32 %[b], %[v11] offset:44\n\t"
                         "s_mov_b64 exec, %[M10]\n\t"
                         "ds_write_b32 %[b], %[v10] offset:40\n\t"
                         "s_mov_b64 exec, %[M9]\n\t"
                         "ds_write_b32 %[b], %[v9] offset:36\n\t"
                         "s_mov_b64 exec, %[M8]\n\t"
                         "ds_write_b32 %[b], %[v8] offset:32\n\t"
                         "s_mov_b64 exec, %[M7]\n\t"
                         "ds_write_b32 %[b], %[v7] offset:28\n\t"
                         "s_mov_b64 exec, %[M6]\n\t"
                         "ds_write_b32 %[b], %[v6] offset:24\n\t"
                         "s_mov_b64 exec, %[M5]\n\t"
                         "ds_write_b32 %[b], %[v5] offset:20\n\t"
                         "s_mov_b64 exec, %[M4]\n\t"
                         "ds_write_b32 %[b], %[v4] offset:16\n\t"
                         "s_mov_b64 exec, %[M3]\n\t"
                         "ds_write_b32 %[b], %[v3] offset:12\n\t"
                         "s_mov_b64 exec, %[M2]\n\t"
                         "ds_write_b32 %[b], %[v2] offset:8\n\t"
                         "s_mov_b64 exec, %[M1]\n\t"
                         "ds_write_b32 %[b], %[v1] offset:4\n\t"
                         "s_mov_b64 exec, %[M0]\n\t"
                         "ds_write_b32 %[b], %[v0]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7), [v8] "=&v"(v8), [v9] "=&v"(v9), [v10] "=&v"(v10), [v11] "=&v"(v11), [v12] "=&v"(v12), [v13] "=&v"(v13), [v14] "=&v"(v14), [v15] "=&v"(v15), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [sv] "=&s"(sv)
                         : [a] "v"(a), [b] "v"(b), [at] "v"(at), [bt] "v"(bt), [M0] "s"(M[0]), [M1] "s"(M[1]), [M2] "s"(M[2]), [M3] "s"(M[3]), [M4] "s"(M[4]), [M5] "s"(M[5]), [M6] "s"(M[6]), [M7] "s"(M[7]), [M8] "s"(M[8]), [M9] "s"(M[9]), [M10] "s"(M[10]), [M11] "s"(M[11]), [M12] "s"(M[12]), [M13] "s"(M[13]), [M14] "s"(M[14]), [M15] "s"(M[15]), [T0] "s"(T[0]), [T1] "s"(T[1]), [T2] "s"(T[2])
                         : "memory");
        }
        // F: matches whose source has left the ring (it lies in the tile's slot in HBM, flushed: copy_match()) and that
        // are at most 64 bytes long, eight at a time: eight byte loads per lane in flight, one wait, eight LDS writes.
        // One by one in stage C each was a memory round trip of its own behind a wait for all stores -- a noisy tile has
        // 5 700 of them with the 8 KiB ring, a third of what was left of the copier's time (decoder_cycle_counters.txt).
        // Their sources are final, they write only their own place: any time before the stages that may read them.
        const bool far_short = is_match && l > 0u && l <= 64u && !near;
        {
            unsigned long long todo = __ballot(far_short);
            if (todo != 0ull)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the flushes this wave has issued are in the slot
            while (todo != 0ull) {
                uint32_t f_dst[8], f_l[8];
                uint8_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    f_l[k] = 0;
                    f_dst[k] = 0;
                    v[k] = 0;
                    if (todo != 0ull) {
                        const int i = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        f_dst[k] = (uint32_t)__builtin_amdgcn_readlane((int)dst, i);
                        f_l[k] = (uint32_t)__builtin_amdgcn_readlane((int)l, i);
                        const uint32_t from = f_dst[k] - (uint32_t)__builtin_amdgcn_readlane((int)dist, i);
                        if ((uint32_t)lane < f_l[k])
                            v[k] = *out_at(o, from + (uint32_t)lane);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((uint32_t)lane < f_l[k])
                        sh.window[(f_dst[k] + (uint32_t)lane) & kWindowMask] = v[k];
            }
        }
        // B2: long matches that read nothing of this sub-batch
        for (unsigned long long todo = __ballot(early_long); todo != 0ull; todo &= todo - 1ull) {
            const int i = __builtin_ctzll(todo);
            copy_long(sh, (uint32_t)__builtin_amdgcn_readlane((int)s_ring, i), (uint32_t)__builtin_amdgcn_readlane((int)d_ring, i),
                      (uint32_t)__builtin_amdgcn_readlane((int)l, i), lane);
        }
        // C: the other matches, in order
        unsigned long long rest = __ballot(is_match && l > 0u && !early && !early_long && !far_short);
        if (diag == 3u)
            rest = 0ull;                // (timing: without the matches carried out one by one)
        const uint32_t pos0 = o.pos;
        while (rest != 0ull) {
            const int i = __builtin_ctzll(rest);
            rest &= rest - 1ull;
            o.pos = (uint32_t)__builtin_amdgcn_readlane((int)dst, i);
            copy_match(sh, o, (uint32_t)__builtin_amdgcn_readlane((int)l, i),
                       (uint32_t)__builtin_amdgcn_readlane((int)dist, i),
                       (uint32_t)__builtin_amdgcn_readlane((int)ahead, i), lane);
        }
        o.pos = pos0 + S < o.limit ? pos0 + S : o.limit;
        done += m;
        if (o.pos - o.flushed >= (uint32_t)kFlush) {
            if (diag == 4u)
                o.flushed += kFlush;    // (timing: without the flush to HBM)
            else
                flush_half(sh, o, lane);
        }
    }
    return 0;
}

// A tile inflate_kernel decodes straight into the destination raster (and untile_kernel leaves alone): see Output.
__device__ __forceinline__ bool tile_in_place(const TileIn &t, uint32_t slot_bytes, const uint8_t *dst, unsigned long long dst_stride)
{
    const uint32_t w = t.chunk_w;
    return !(t.flags & (GCN10_TILE_RAW | GCN10_TILE_PREDICTOR2)) && w >= 16u && (w & (w - 1u)) == 0u && t.src_x == 0u &&
           t.src_y == 0u && t.copy_w == w && t.copy_h > 0u && (unsigned long long)t.copy_h * w == t.out_len &&
           t.out_len <= slot_bytes && ((reinterpret_cast<uintptr_t>(dst) | t.dst_off | dst_stride) & 15u) == 0u;
}

// One workgroup of two wavefronts per stream: wave 0 decodes bits into tokens, wave 1 carries
// the tokens out; they swap halves of a small token ring at a barrier every kBatch tokens.
// diag (gcn10_gpu_set_option "inflate_diag", timing experiments only, output invalid; 3 = the copier skips the
// matches it carries out one by one, 4 = it skips its flushes to HBM): 1 = the copier
// carries nothing out, 2 = the decoder hands over empty batches after decoding them
__global__ __launch_bounds__(128) void inflate_kernel(const uint8_t *comp, const TileIn *tiles, uint32_t n_tiles,
                                                      uint8_t *scratch, uint32_t slot_bytes, uint32_t *status,
                                                      uint32_t diag, uint8_t *dst, unsigned long long dst_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Shared &sh = *reinterpret_cast<Shared *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles)
        return;
    const TileIn tin = tiles[tile];
    if (tin.flags & GCN10_TILE_RAW) {
        // not a zlib stream: the chunk's bytes lie in `comp` as they are; untile_kernel copies the window
        if (threadIdx.x == 0)
            status[tile] = tin.in_len >= tin.out_len &&
                                   window_ok(tin.out_len, tin.chunk_w, tin.src_x, tin.src_y, tin.copy_w, tin.copy_h,
                                             0xffffffffu)
                               ? 0u
                               : (uint32_t)kErrWindow;
        return;
    }

    Decoder d;
    d.r.in = reinterpret_cast<const uint32_t *>(comp + tin.in_off);
    d.r.n_dwords = (tin.in_len + 3u) / 4u;
    d.r.lane = lane;
    d.in_len = tin.in_len;
    d.state = kNeedHeader;
    d.P = 0;
    d.pos = 0;
    d.limit = tin.out_len < slot_bytes ? tin.out_len : slot_bytes;
    d.stored_left = d.stored_at = 0;
    d.err = kOk;
    d.last = false;
    Output o;
    o.out = scratch + (size_t)tile * slot_bytes;
    o.wshift = o.wmask = 0;
    o.stride = 0;
    if (tile_in_place(tin, slot_bytes, dst, dst_stride)) {
        o.out = dst + tin.dst_off;
        o.wshift = 31u - (uint32_t)__builtin_clz(tin.chunk_w);
        o.wmask = tin.chunk_w - 1u;
        o.stride = dst_stride;
    }
    o.limit = d.limit;
    o.pos = 0;
    o.flushed = 0;

    if (wave == 0) {
        reader_seek(d.r, 0);
        refill(d.r);
        const uint32_t cmf = take_bits(d.r, 8), flg = take_bits(d.r, 8);
        if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 0x20u) || ((cmf << 8 | flg) % 31u) != 0u) {
            d.err = kErrHeader;
            d.state = kDone;
        }
        if (lane == 0) {
            sh.stop = 0xffffffffu;
            sh.err = kOk;
            sh.count[0] = sh.count[1] = 0;
        }
    }
    __syncthreads();
    bool announced = false;
    uint32_t c_err = 0;
    for (uint32_t it = 0;; it++) {
        const uint32_t cur = it & 1u;
        if (wave == 0) {
            if (!announced) {
                const uint32_t n = decode_batch(sh, d, sh.ring[cur], lane);
                if (lane == 0)
                    sh.count[cur] = diag == 2u ? 0u : n;
                if (d.state == (uint32_t)kDone) {
                    if (!d.err && reader_byte_pos(d.r) > d.in_len + 4u)
                        d.err = kErrInput;
                    if (lane == 0) {
                        sh.stop = it + 1u;          // the copier still has this batch to carry out
                        sh.err = d.err;
                    }
                    announced = true;
                }
            }
        }
        else if (it > 0 && c_err == 0) {
            if (diag != 1u)
                c_err = copy_batch(sh, o, sh.ring[cur ^ 1u], uniform(sh.count[cur ^ 1u]), comp + tin.in_off, lane, diag);
        }
        __syncthreads();
        if (it >= uniform(sh.stop))
            break;
    }
    if (wave == 0)
        return;
    uint32_t err = c_err ? c_err : uniform(sh.err);
    if (!err && !window_ok(tin.out_len, tin.chunk_w, tin.src_x, tin.src_y, tin.copy_w, tin.copy_h, slot_bytes))
        err = kErrWindow;

    // what is still in the window, then zeros up to the tile's size (as a short stream reads on the host)
    {
        const uint32_t end = o.pos < o.limit ? o.pos : o.limit;
        for (uint32_t i = o.flushed + (uint32_t)lane; i < end; i += 64)
            *out_at(o, i) = sh.window[i & kWindowMask];
        for (uint32_t i = end + (uint32_t)lane; i < tin.out_len && i < slot_bytes; i += 64)
            *out_at(o, i) = 0;
    }
    if (lane == 0)
        status[tile] = err;
}

// The wanted window of every decoded tile -> the row-major landcover block.  A raw tile
// (GCN10_TILE_RAW: TIFF Compression 1) is read where it lies among the staged bytes; a tile
// written with TIFF Predictor 2 (horizontal differencing, per chunk row) is summed back on the
// way: byte x of a row = sum of the stored bytes 0..x of that row, modulo 256 -- the inverse the
// host reader applies in tiff.c, decode_chunk().
__global__ __launch_bounds__(256) void untile_kernel(const TileIn *tiles, const uint8_t *scratch, uint32_t slot_bytes,
                                                     const uint8_t *comp, uint8_t *dst, unsigned long long dst_stride)
{
    typedef uint32_t u32_u __attribute__((aligned(1)));
    const TileIn tin = tiles[blockIdx.x];
    const bool raw = (tin.flags & GCN10_TILE_RAW) != 0;
    if (!window_ok(tin.out_len, tin.chunk_w, tin.src_x, tin.src_y, tin.copy_w, tin.copy_h, raw ? 0xffffffffu : slot_bytes) ||
        (raw && tin.in_len < tin.out_len))
        return;                                 // inflate_kernel has set the status
    if (tile_in_place(tin, slot_bytes, dst, dst_stride))
        return;                                 // inflate_kernel has written its rows where they belong
    const uint8_t *chunk = raw ? comp + tin.in_off : scratch + (size_t)blockIdx.x * slot_bytes;
    uint8_t *out = dst + tin.dst_off;
    const int lane = (int)(threadIdx.x & 63u);
    if (tin.flags & GCN10_TILE_PREDICTOR2) {
        // one wave per row, 1024 stored bytes per trip: 16 per lane summed in the lane, the lanes' totals
        // by a wave scan, the carry of the row's earlier trips on top
        const uint32_t end = tin.src_x + tin.copy_w;            // bytes of the row that matter
        for (uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6); y < tin.copy_h; y += gridDim.y * 4u) {
            const uint8_t *s = chunk + (size_t)(tin.src_y + y) * tin.chunk_w;
            uint8_t *d = out + (size_t)y * dst_stride;
            uint32_t carry = 0;
            for (uint32_t x0 = 0; x0 < end; x0 += 1024u) {
                const uint32_t x = x0 + (uint32_t)lane * 16u;
                uint8_t v[16];
                uint32_t acc = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    acc += x + (uint32_t)k < end ? s[x + (uint32_t)k] : 0u;
                    v[k] = (uint8_t)acc;
                }
                uint32_t incl = acc & 0xffu;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = __shfl_up(incl, off, 64);
                    if (lane >= off)
                        incl += up;
                }
                const uint32_t before = carry + incl - (acc & 0xffu);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t xx = x + (uint32_t)k;
                    if (xx >= tin.src_x && xx < end)
                        d[xx - tin.src_x] = (uint8_t)(v[k] + before);
                }
                carry = (carry + (uint32_t)__shfl(incl, 63, 64)) & 0xffu;
            }
        }
        return;
    }
    const uint8_t *src = chunk + (size_t)tin.src_y * tin.chunk_w + tin.src_x;
    // rows whose source and destination are 16-byte aligned (interior tiles of a window that starts on a multiple
    // of 16 pixels): 16 bytes per lane, a 1024-pixel row in one wave instruction each way
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out) | tin.chunk_w | dst_stride | tin.copy_w) & 15u) == 0u) {
        const uint32_t w16 = tin.copy_w / 16u;
        for (uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6); y < tin.copy_h; y += gridDim.y * 4u) {
            const u32x4 *s = reinterpret_cast<const u32x4 *>(src + (size_t)y * tin.chunk_w);
            u32x4 *d = reinterpret_cast<u32x4 *>(out + (size_t)y * dst_stride);
            for (uint32_t i = threadIdx.x & 63u; i < w16; i += 64u)
                __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
        }
        return;
    }
    const uint32_t w4 = tin.copy_w / 4u;
    for (uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6); y < tin.copy_h; y += gridDim.y * 4u) {
        const uint8_t *s = src + (size_t)y * tin.chunk_w;
        uint8_t *d = out + (size_t)y * dst_stride;
        for (uint32_t i = threadIdx.x & 63u; i < w4; i += 64u)
            reinterpret_cast<u32_u *>(d)[i] = reinterpret_cast<const u32_u *>(s)[i];
        for (uint32_t i = w4 * 4u + (threadIdx.x & 63u); i < tin.copy_w; i += 64u)
            d[i] = s[i];
    }
}

}  // namespace

extern "C" {

int gcn10_gpu_inflate_tiles(gcn10_gpu_ctx *ctx, const uint8_t *comp_dev, const gcn10_inflate_tile *tiles_dev,
                            int n_tiles, uint32_t chunk_bytes, uint8_t *dst_dev, size_t dst_stride,
                            uint32_t *status_dev, gcn10_stream_t stream)
{
    int rc = use_device(ctx);
    if (rc)
        return rc;
    if (n_tiles < 0 || chunk_bytes == 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_inflate_tiles: bad shape (%d tiles of %u bytes)", n_tiles, chunk_bytes);
    if (n_tiles == 0)
        return GCN10_OK;
    if (!comp_dev || !tiles_dev || !dst_dev || !status_dev)
        return fail(GCN10_E_INVAL, "gcn10_gpu_inflate_tiles: null pointer");
    if ((reinterpret_cast<uintptr_t>(comp_dev) & 15u) != 0)
        return fail(GCN10_E_INVAL, "gcn10_gpu_inflate_tiles: compressed bytes must be 16-byte aligned");
    const uint32_t slot = (chunk_bytes + 255u) & ~255u;
    const size_t need = (size_t)slot * (size_t)n_tiles;
    if (need > ctx->inflate_ws_cap) {
        HIP_TRY(hipDeviceSynchronize());        // the old workspace may still be in use
        if (ctx->inflate_ws)
            HIP_TRY(hipFree(ctx->inflate_ws));
        ctx->inflate_ws = nullptr;
        ctx->inflate_ws_cap = 0;
        HIP_TRY(hipMalloc(&ctx->inflate_ws, need));
        ctx->inflate_ws_cap = need;
    }
    static_assert(kWindow / 2 + kWindow / 4 + 258 <= kWindow && (kWindow & (kWindow - 1)) == 0 && kWindow >= 4096, "window invariant");
    static_assert(sizeof(Shared) <= (kWindow == 8192 ? 16 : 26) * 1024, "ten (8 KiB ring) or six (16 KiB) streams per CU of 160 KiB LDS");
    hipStream_t s = as_stream(ctx, stream);
    uint8_t *scratch = reinterpret_cast<uint8_t *>(ctx->inflate_ws);
    hipLaunchKernelGGL(inflate_kernel, dim3((uint32_t)n_tiles), dim3(128), sizeof(Shared), s, comp_dev,
                       reinterpret_cast<const TileIn *>(tiles_dev), (uint32_t)n_tiles, scratch, slot, status_dev,
                       (uint32_t)ctx->inflate_diag, dst_dev, (unsigned long long)dst_stride);
    hipLaunchKernelGGL(untile_kernel, dim3((uint32_t)n_tiles, 16), dim3(256), 0, s,
                       reinterpret_cast<const TileIn *>(tiles_dev), scratch, slot, comp_dev, dst_dev,
                       (unsigned long long)dst_stride);
    HIP_TRY(hipGetLastError());
    return GCN10_OK;
}

}  // extern "C"

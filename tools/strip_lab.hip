// strip_lab.hip -- same-box A/B bench of single-raster CN strip kernel variants (BASELINE config 2)
// next to plain copy kernels of the same launch shape.  Diagnostic only: the winner is ported into
// gcn10_amd/csrc/gcn10_gpu.hip, where the parity tests check it against the oracle.  Every variant's
// raster is compared on the device with a byte-per-thread kernel of the plain formula.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/strip_lab tools/strip_lab.hip
//   tools/strip_lab [W=36000] [rows=36000] [reps=20] > gpurun_out/strip_lab.jsonl
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int kPlane1 = 260;            // byte LUT [6][260]
constexpr int kLutA = 6 * kPlane1;
constexpr int kLutB = 65536;            // lane-replicated LUT [256 classes][256]: byte (s&3)|(rep<<2)|((s>>2)<<7)

struct P {
    const uint8_t *esa;
    const uint8_t *hx;      // soil codes, x-expanded, 16 B of padding in front and behind every row
    const int32_t *cj;
    const uint8_t *lut;     // global image of the LDS table
    uint8_t *out;
    uint32_t W, rows, npix, nvec, hx_stride, hx_rows, ntrips;
};

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

__device__ __forceinline__ int32_t sload_i32(const int32_t *base, uint32_t i)
{
    typedef const int32_t __attribute__((address_space(4))) *cp_t;
    cp_t cp = (cp_t)(uintptr_t)base;
    return cp[__builtin_amdgcn_readfirstlane(i)];
}

__device__ __forceinline__ u32x4 load16_any(const uint8_t *p)
{
    typedef u32x4 u32x4_u __attribute__((aligned(1)));
    return *reinterpret_cast<const u32x4_u *>(p);
}

// ---------------------------------------------------------------------------------------------
// reference: one thread per pixel, plain formula (hx in nibble encoding)
// ---------------------------------------------------------------------------------------------
__global__ void ref_kernel(const uint8_t *esa, const uint8_t *hxA, uint32_t hx_stride, const int32_t *cj,
                           const uint8_t *lutA, uint8_t *out, uint32_t W, uint32_t npix)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t y = i / W, x = i - y * W;
        uint32_t cd = hxA[(size_t)cj[y] * hx_stride + x] & 0xf;
        out[i] = lutA[cd * kPlane1 + esa[i]];
    }
}

__global__ void diff_kernel(const u32x4 *a, const u32x4 *b, size_t nvec, unsigned long long *count)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 x = a[i] ^ b[i];
        c += (x[0] | x[1] | x[2] | x[3]) != 0;
    }
    if (c)
        atomicAdd(count, c);
}

// ---------------------------------------------------------------------------------------------
// the variant kernel
//   LUTK  0: byte LUT [6][260] (nibble soil codes), 2: lane-replicated conflict-free LUT (64 KB)
//   ILP   1024-px wave chunks per trip, all loads of a trip issued before any is used
//   PF    software pipeline: the next trip's loads are issued before this trip's gathers and stores
//         (two register sets used alternately, no copies)
//   DIAG  bit 0: no table gather, bit 1: no soil load (timing only)
// ---------------------------------------------------------------------------------------------
template <int ILP>
struct Trip {
    u32x4 e[ILP], c[ILP];
    uint32_t i0[ILP];
};

template <int LUTK, int ILP, int THREADS, int DIAG, int NTM>
__device__ __forceinline__ void issue(const P &p, uint32_t trip, uint32_t lane16, uint32_t wave, Trip<ILP> &t)
{
    uint32_t x[ILP], r0[ILP], r1[ILP], wb[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        wb[u] = __builtin_amdgcn_readfirstlane((trip * ILP + u) * (uint32_t)(THREADS * 16) + wave * 1024u);
        const uint32_t wbc = wb[u] < p.npix ? wb[u] : 0u;
        const uint32_t y = wbc / p.W;
        x[u] = wbc - y * p.W;
        r0[u] = (uint32_t)sload_i32(p.cj, y);
        r1[u] = (uint32_t)sload_i32(p.cj, y + 1u < p.rows ? y + 1u : y);
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        t.i0[u] = wb[u] + lane16;
        const bool live = t.i0[u] < p.nvec * 16u;
        const u32x4 *pe = reinterpret_cast<const u32x4 *>(p.esa + (live ? t.i0[u] : 0u));
        t.e[u] = (NTM & 1) ? __builtin_nontemporal_load(pe) : *pe;
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        uint32_t xl = x[u] + lane16;
        const bool wrap = xl >= p.W;
        xl = wrap ? xl - p.W : xl;
        uint32_t row = wrap ? r1[u] : r0[u];
        row = row < p.hx_rows ? row : p.hx_rows - 1u;
        if (DIAG & 2) {
            t.c[u] = u32x4{row, xl, 0u, 0u} & 0x01010101u;
            continue;
        }
        if (DIAG & 4) {
            // timing only: one dword per 16-px group {code a, code b, split} instead of 16 code bytes
            const uint32_t w = *reinterpret_cast<const uint32_t *>(p.hx + (size_t)row * (p.hx_stride & ~15u) + (xl >> 2));
            t.c[u] = u32x4{w, 0u, 0u, 0u};
            continue;
        }
        const uint8_t *pa = p.hx + (size_t)row * p.hx_stride + xl;
        u32x4 a = load16_any(pa);
        if (p.W & 15u) {
            // the lane whose 16 pixels cross a row end takes its last bytes from the next row's soil row:
            // a second load that starts n bytes before that row (16 B of padding are in front of row 0)
            const uint32_t n = p.W - xl;                // bytes of this lane that are still in row y
            const bool strad = !wrap && n < 16u;
            if (__builtin_amdgcn_ballot_w64(strad) != 0ull) {
                uint32_t rb = r1[u] < p.hx_rows ? r1[u] : p.hx_rows - 1u;
                const uint8_t *pb = strad ? p.hx + (size_t)rb * p.hx_stride - n : pa;
                const u32x4 b = load16_any(pb);
                const uint32_t nn = strad ? n : 16u;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int32_t m = (int32_t)nn - 4 * j;      // bytes of dword j taken from a
                    const uint32_t mask = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : (1u << (8 * m)) - 1u);
                    a[j] = (a[j] & mask) | (b[j] & ~mask);
                }
            }
        }
        t.c[u] = a;
    }
}

template <int LUTK, int ILP, int DIAG, int NTM>
__device__ __forceinline__ void finish(const P &p, const uint8_t *lut, uint32_t lane_rep, const Trip<ILP> &t)
{
#pragma unroll
    for (int u = 0; u < ILP; u++) {
        u32x4 v;
        u32x4 cexp = t.c[u];
        if (DIAG & 4) {
            const uint32_t w = t.c[u][0];
            const uint32_t a = (w & 0xffu) * 0x01010101u, b = ((w >> 8) & 0xffu) * 0x01010101u;
            const int32_t split = (int32_t)((w >> 16) & 0x1fu);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int32_t m = split - 4 * j;
                const uint32_t mask = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : (1u << (8 * m)) - 1u);
                cexp[j] = (a & mask) | (b & ~mask);
            }
        }
        if (DIAG & 1) {
            v = t.e[u] ^ cexp;
        }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = t.e[u][j];
                const uint32_t cd = cexp[j];
                uint32_t b[4];
                if (LUTK == 2) {
                    const uint32_t m = (cd & 0x83838383u) | lane_rep;
                    b[0] = lut[perm(e, m, 0x0c0c0400u)];
                    b[1] = lut[perm(e, m, 0x0c0c0501u)];
                    b[2] = lut[perm(e, m, 0x0c0c0602u)];
                    b[3] = lut[perm(e, m, 0x0c0c0703u)];
                }
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lc = (e >> (8 * q)) & 0xffu;
                        const uint32_t s = (cd >> (8 * q)) & 0xfu;
                        b[q] = lut[s * (uint32_t)kPlane1 + lc];
                    }
                }
                v[j] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            }
        }
        if (t.i0[u] < p.nvec * 16u) {
            if (NTM & 2)
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p.out + t.i0[u]));
            else
                *reinterpret_cast<u32x4 *>(p.out + t.i0[u]) = v;
        }
    }
}

template <int LUTK, int ILP, bool PF, int THREADS, int DIAG, int MAP = 1, int NTM = 3>
__global__ __launch_bounds__(THREADS) void lab_kernel(const P p)
{
    constexpr int kLutBytes = LUTK == 2 ? kLutB : ((kLutA + 15) & ~15);
    __shared__ __attribute__((aligned(16))) uint8_t lut[kLutBytes];
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.lut);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lut);
        for (int i = threadIdx.x; i < kLutBytes / 16; i += THREADS)
            dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane16 = lane * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane_rep = (lane & 31u) * 0x04040404u;

    // every XCD (blockIdx % 8) streams a contiguous eighth of the trips
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    if (MAP == 2) {
        // region-synchronous sub-slabs: the strip is cut at multiples of 64 MiB of the landcover's ADDRESS;
        // every region is split into eight sub-slabs, one per XCD, and all XCDs sweep one region at a time,
        // so that at any moment all reads fall into one 64 MiB-aligned piece of memory
        constexpr uint32_t kTripBytes = THREADS * 16 * ILP;
        constexpr uint32_t kRegionTrips = (64u << 20) / kTripBytes, kSubTrips = kRegionTrips / 8u;
        const uint32_t phase = (uint32_t)(((uintptr_t)p.esa & ((64u << 20) - 1u)) / kTripBytes);
        const uint32_t xcd = b & 7u, first = b >> 3, step = nb / 8u;
        const uint32_t nregions = (p.ntrips + phase + kRegionTrips - 1u) / kRegionTrips;
        for (uint32_t r = 0; r < nregions; r++)
            for (uint32_t t = first; t < kSubTrips; t += step) {
                const uint32_t g = r * kRegionTrips + xcd * kSubTrips + t;
                if (g < phase || g - phase >= p.ntrips)
                    continue;
                Trip<ILP> tr;
                issue<LUTK, ILP, THREADS, DIAG, NTM>(p, g - phase, lane16, wave, tr);
                finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, tr);
            }
    }
    else {
    const uint32_t per = (p.ntrips + 7u) / 8u;
    const uint32_t lo = (b & 7u) * per;
    const uint32_t end = MAP ? (lo + per < p.ntrips ? lo + per : p.ntrips) : p.ntrips;
    const uint32_t step = MAP ? nb / 8u : nb;
    uint32_t trip = MAP ? lo + (b >> 3) : b;

    if (!PF) {
        for (; trip < end; trip += step) {
            Trip<ILP> t;
            issue<LUTK, ILP, THREADS, DIAG, NTM>(p, trip, lane16, wave, t);
            finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, t);
        }
    }
    else if (trip < end) {
        Trip<ILP> ta, tb;
        issue<LUTK, ILP, THREADS, DIAG, NTM>(p, trip, lane16, wave, ta);
        for (;;) {
            trip += step;
            if (trip >= end) {
                finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, ta);
                break;
            }
            issue<LUTK, ILP, THREADS, DIAG, NTM>(p, trip, lane16, wave, tb);
            finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, ta);
            trip += step;
            if (trip >= end) {
                finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, tb);
                break;
            }
            issue<LUTK, ILP, THREADS, DIAG, NTM>(p, trip, lane16, wave, ta);
            finish<LUTK, ILP, DIAG, NTM>(p, lut, lane_rep, tb);
        }
    }
    }
    // the last npix % 16 pixels, byte-wise
    if (blockIdx.x == 0 && threadIdx.x < (p.npix & 15u) && !(DIAG & 3)) {
        const uint32_t i = p.nvec * 16u + threadIdx.x;
        const uint32_t y = i / p.W, x = i - y * p.W;
        uint32_t row = (uint32_t)p.cj[y];
        row = row < p.hx_rows ? row : p.hx_rows - 1u;
        const uint32_t cd = p.hx[(size_t)row * p.hx_stride + x];
        if (LUTK == 2)
            p.out[i] = lut[((uint32_t)p.esa[i] << 8) | (cd & 0x83u) | (lane_rep & 0xffu)];
        else
            p.out[i] = lut[(cd & 0xfu) * (uint32_t)kPlane1 + p.esa[i]];
    }
}

// ---------------------------------------------------------------------------------------------
// plain 1R:1W copy of the same launch shape (XCD slabs)
// ---------------------------------------------------------------------------------------------
template <int UN, int THREADS, int MAP = 1>
__global__ __launch_bounds__(THREADS) void copy_kernel(const u32x4 *in, u32x4 *out, size_t nvec)
{
    const size_t nchunk = (nvec + THREADS * UN - 1) / (THREADS * UN);
    if (MAP == 2) {
        constexpr uint32_t kChunkBytes = THREADS * 16 * UN;
        constexpr uint32_t kRegion = (64u << 20) / kChunkBytes, kSub = kRegion / 8u;
        const uint32_t phase = (uint32_t)(((uintptr_t)in & ((64u << 20) - 1u)) / kChunkBytes);
        const uint32_t xcd = blockIdx.x & 7u, first = blockIdx.x >> 3, step = gridDim.x / 8u;
        const uint32_t nregions = (uint32_t)((nchunk + phase + kRegion - 1u) / kRegion);
        for (uint32_t r = 0; r < nregions; r++)
            for (uint32_t t = first; t < kSub; t += step) {
                const uint32_t g = r * kRegion + xcd * kSub + t;
                if (g < phase || g - phase >= nchunk)
                    continue;
                const size_t c = g - phase;
                u32x4 v[UN];
                size_t idx[UN];
#pragma unroll
                for (int u = 0; u < UN; u++) {
                    idx[u] = (c * UN + u) * THREADS + threadIdx.x;
                    v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0));
                }
#pragma unroll
                for (int u = 0; u < UN; u++)
                    if (idx[u] < nvec)
                        __builtin_nontemporal_store(v[u], out + idx[u]);
            }
        return;
    }
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = MAP ? xcd * per + (blockIdx.x >> 3) : blockIdx.x;
    const size_t cend = MAP ? ((xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk) : nchunk;
    const size_t cstep = MAP ? gridDim.x / 8 : gridDim.x;
    for (; c < cend; c += cstep) {
        u32x4 v[UN];
        size_t idx[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            idx[u] = (c * UN + u) * THREADS + threadIdx.x;
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0));
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
    }
}

// ---------------------------------------------------------------------------------------------
// phase-gated copy: every wave issues its loads only while bit TBITS of the chip-wide 100 MHz
// counter (s_memrealtime) is 0 and its stores only while it is 1, so that the whole chip alternates
// between reading and writing HBM in slices of 2^TBITS * 10 ns without any communication.  Tests the
// guess that the ~10 % a 1R:1W stream loses against read-only + write-only time sharing is bus
// turnaround between the two directions.
// ---------------------------------------------------------------------------------------------
template <int TBITS>
__device__ __forceinline__ void wait_phase(uint32_t want)
{
    while (((uint32_t)(__builtin_amdgcn_s_memrealtime() >> TBITS) & 1u) != want)
        __builtin_amdgcn_s_sleep(1);
}

template <int UN, int THREADS, int TBITS>
__global__ __launch_bounds__(THREADS) void copy_phased_kernel(const u32x4 *in, u32x4 *out, size_t nvec)
{
    const size_t nchunk = (nvec + THREADS * UN - 1) / (THREADS * UN);
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = xcd * per + (blockIdx.x >> 3);
    const size_t cend = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk;
    const size_t cstep = gridDim.x / 8;
    for (; c < cend; c += cstep) {
        u32x4 v[UN];
        size_t idx[UN];
        wait_phase<TBITS>(0u);
#pragma unroll
        for (int u = 0; u < UN; u++) {
            idx[u] = (c * UN + u) * THREADS + threadIdx.x;
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wait_phase<TBITS>(1u);
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
    }
}

// ---------------------------------------------------------------------------------------------
// copy with other cache policies on the store: STM 1 = sc1, 2 = sc0 sc1 (write-through, the line is
// dropped from L2), 3 = nt sc1, 4 = plain; loads LDM 0 = nt, 1 = plain, 2 = sc1
// ---------------------------------------------------------------------------------------------
template <int STM>
__device__ __forceinline__ void store_policy(u32x4 *p, u32x4 v)
{
    if (STM == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (STM == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (STM == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else *p = v;
}

template <int LDM, int STM>
__global__ __launch_bounds__(256) void copy_policy_kernel(const u32x4 *in, u32x4 *out, size_t nvec)
{
    const size_t nchunk = (nvec + 511) / 512;
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = xcd * per + (blockIdx.x >> 3);
    const size_t cend = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk;
    const size_t cstep = gridDim.x / 8;
    for (; c < cend; c += cstep) {
        u32x4 v[2];
        size_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (c * 2 + u) * 256 + threadIdx.x;
            const u32x4 *p = in + (idx[u] < nvec ? idx[u] : 0);
            if (LDM == 0) v[u] = __builtin_nontemporal_load(p);
            else if (LDM == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
            else v[u] = *p;
        }
        if (LDM == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (idx[u] < nvec)
                store_policy<STM>(out + idx[u], v[u]);
    }
}

// ---------------------------------------------------------------------------------------------
// where does the strip kernel lose against the plain copy?  The copy loop (two chunks per trip, XCD
// slabs) with the strip kernel's ingredients added one at a time:
//   LEVEL 1  + the per-trip scalar division and the two scalar loads of cj (results kept alive)
//   LEVEL 2  + the soil stream (one 16-byte load per chunk from the x-expanded rows, L2 hits)
//   LEVEL 3  + the table gather from LDS (the real raster)
// ---------------------------------------------------------------------------------------------
template <int LEVEL>
__global__ __launch_bounds__(256) void copy_plus_kernel(const P p)
{
    __shared__ __attribute__((aligned(16))) uint8_t lut[(kLutA + 15) & ~15];
    if (LEVEL >= 3) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.lut);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lut);
        for (int i = threadIdx.x; i < (int)sizeof(lut) / 16; i += 256)
            dst[i] = src[i];
        __syncthreads();
    }
    const u32x4 *in = reinterpret_cast<const u32x4 *>(p.esa);
    u32x4 *out = reinterpret_cast<u32x4 *>(p.out);
    const uint32_t nvec = p.nvec;
    const uint32_t nchunk = (nvec + 511u) / 512u;
    const uint32_t per = (nchunk + 7u) / 8u, xcd = blockIdx.x & 7u;
    uint32_t c = xcd * per + (blockIdx.x >> 3);
    const uint32_t cend = (xcd + 1u) * per < nchunk ? (xcd + 1u) * per : nchunk, cstep = gridDim.x / 8u;
    const uint32_t lane16 = (threadIdx.x & 63u) * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (; c < cend; c += cstep) {
        u32x4 v[2], h[2];
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (c * 2u + u) * 256u + threadIdx.x;
            uint32_t row = 0, xl = 0;
            if (LEVEL >= 1) {
                const uint32_t wb = __builtin_amdgcn_readfirstlane(((c * 2u + u) * 256u + wave * 64u) * 16u);
                const uint32_t wbc = wb < p.npix ? wb : 0u;
                const uint32_t y = wbc / p.W;
                const uint32_t x = wbc - y * p.W;
                const uint32_t r0 = (uint32_t)sload_i32(p.cj, y);
                const uint32_t r1 = (uint32_t)sload_i32(p.cj, y + 1u < p.rows ? y + 1u : y);
                xl = x + lane16;
                const bool wrap = xl >= p.W;
                xl = wrap ? xl - p.W : xl;
                row = wrap ? r1 : r0;
                row = row < p.hx_rows ? row : p.hx_rows - 1u;
            }
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0u));
            if (LEVEL == 4) {           // timing only: soil codes packed two per byte, one 8-byte load per lane
                typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
                const u32x2_u t = *reinterpret_cast<const u32x2_u *>(p.hx + (size_t)row * p.hx_stride + (xl >> 1));
                h[u] = u32x4{t[0], t[1], 0u, 0u};
            }
            else if (LEVEL == 5) {      // timing only: four per byte, one 4-byte load per lane
                typedef uint32_t u32_u __attribute__((aligned(1)));
                h[u] = u32x4{*reinterpret_cast<const u32_u *>(p.hx + (size_t)row * p.hx_stride + (xl >> 2)), 0u, 0u, 0u};
            }
            else if (LEVEL >= 2)
                h[u] = load16_any(p.hx + (size_t)row * p.hx_stride + xl);
            else
                h[u] = u32x4{row, xl, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            u32x4 o;
            if (LEVEL == 3) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t b[4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        b[q] = lut[((h[u][j] >> (8 * q)) & 0xfu) * (uint32_t)kPlane1 + ((v[u][j] >> (8 * q)) & 0xffu)];
                    o[j] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                }
            }
            else {
                o = v[u] ^ (h[u] & 0u);       // keeps the address work and the soil load alive, leaves the copy a copy
                if (LEVEL >= 1 && (h[u][0] | h[u][1] | h[u][2] | h[u][3]) == 0xdeadbeefu)
                    o[0] ^= 1u;
            }
            if (idx[u] < nvec)
                __builtin_nontemporal_store(o, out + idx[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: every variant is a closure that launches one kernel with the events attached to the
// dispatch; variants are timed in interleaved rounds (A B C ... A B C ...) so that drift of the box
// (clocks, temperature) falls on all of them alike.
// ---------------------------------------------------------------------------------------------
#include <functional>
#include <string>

struct Variant {
    std::string name;
    int blocks, threads;
    double bytes;
    std::function<void(hipEvent_t, hipEvent_t)> launch;
    uint8_t *check;             // raster to compare with the reference after the first launch, or null
    std::vector<float> ms;
    long long bad = -1;
};

struct Lab {
    P pA, pB;
    uint8_t *ref;
    uint8_t *outs[3];
    unsigned long long *d_count;
    int cus;
    hipStream_t s;
    std::vector<Variant> v;
};

template <int LUTK, int ILP, bool PF, int THREADS, int DIAG, int MAP = 1, int NTM = 3>
static void add_variant(Lab &L, int wg_per_cu, int out_idx = 0)
{
    P p = LUTK == 2 ? L.pB : L.pA;
    p.out = L.outs[out_idx];
    const uint32_t trip_px = THREADS * 16 * ILP;
    p.ntrips = (p.npix + trip_px - 1) / trip_px;
    int blocks = L.cus * wg_per_cu;
    blocks -= blocks % 8;
    char name[160];
    snprintf(name, sizeof name, "lut%d_ilp%d_pf%d_t%d_diag%d_map%d_nt%d_wg%d_out%d", LUTK, ILP, (int)PF, THREADS, DIAG, MAP,
             NTM, wg_per_cu, out_idx);
    Variant v;
    v.name = name; v.blocks = blocks; v.threads = THREADS;
    v.bytes = 2.0 * p.npix + 1440.0 * 1440.0 + 4.0 * (p.W + p.rows);
    v.check = DIAG ? nullptr : p.out;
    hipStream_t s = L.s;
    v.launch = [p, blocks, s](hipEvent_t a, hipEvent_t b) {
        P pp = p;
        void *args[] = {&pp};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(lab_kernel<LUTK, ILP, PF, THREADS, DIAG, MAP, NTM>), dim3(blocks),
                              dim3(THREADS), args, 0, s, a, b, 0));
    };
    L.v.push_back(v);
}

template <int UN, int THREADS, int MAP = 1>
static void add_copy(Lab &L, int wg_per_cu)
{
    int blocks = L.cus * wg_per_cu;
    blocks -= blocks % 8;
    char name[64];
    snprintf(name, sizeof name, "copy_un%d_t%d_map%d_wg%d", UN, THREADS, MAP, wg_per_cu);
    const u32x4 *in = (const u32x4 *)L.pA.esa;
    u32x4 *out = (u32x4 *)L.outs[0];
    size_t nvec = L.pA.npix / 16;
    Variant v;
    v.name = name; v.blocks = blocks; v.threads = THREADS; v.bytes = 2.0 * L.pA.npix; v.check = nullptr;
    hipStream_t s = L.s;
    v.launch = [in, out, nvec, blocks, s](hipEvent_t a, hipEvent_t b) {
        const u32x4 *i_ = in; u32x4 *o_ = out; size_t n_ = nvec;
        void *args[] = {&i_, &o_, &n_};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(copy_kernel<UN, THREADS, MAP>), dim3(blocks), dim3(THREADS), args, 0,
                              s, a, b, 0));
    };
    L.v.push_back(v);
}

template <int UN, int THREADS, int TBITS>
static void add_phased(Lab &L, int wg_per_cu)
{
    int blocks = L.cus * wg_per_cu;
    blocks -= blocks % 8;
    char name[64];
    snprintf(name, sizeof name, "copy_phased_un%d_t%d_tbits%d_wg%d", UN, THREADS, TBITS, wg_per_cu);
    const u32x4 *in = (const u32x4 *)L.pA.esa;
    u32x4 *out = (u32x4 *)L.outs[0];
    size_t nvec = L.pA.npix / 16;
    Variant v;
    v.name = name; v.blocks = blocks; v.threads = THREADS; v.bytes = 2.0 * L.pA.npix; v.check = nullptr;
    hipStream_t s = L.s;
    v.launch = [in, out, nvec, blocks, s](hipEvent_t a, hipEvent_t b) {
        const u32x4 *i_ = in; u32x4 *o_ = out; size_t n_ = nvec;
        void *args[] = {&i_, &o_, &n_};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(copy_phased_kernel<UN, THREADS, TBITS>), dim3(blocks), dim3(THREADS),
                              args, 0, s, a, b, 0));
    };
    L.v.push_back(v);
}

template <int LDM, int STM>
static void add_policy(Lab &L, int wg_per_cu)
{
    int blocks = L.cus * wg_per_cu;
    blocks -= blocks % 8;
    char name[64];
    snprintf(name, sizeof name, "copy_policy_ld%d_st%d_wg%d", LDM, STM, wg_per_cu);
    const u32x4 *in = (const u32x4 *)L.pA.esa;
    u32x4 *out = (u32x4 *)L.outs[0];
    size_t nvec = L.pA.npix / 16;
    Variant v;
    v.name = name; v.blocks = blocks; v.threads = 256; v.bytes = 2.0 * L.pA.npix; v.check = nullptr;
    hipStream_t s = L.s;
    v.launch = [in, out, nvec, blocks, s](hipEvent_t a, hipEvent_t b) {
        const u32x4 *i_ = in; u32x4 *o_ = out; size_t n_ = nvec;
        void *args[] = {&i_, &o_, &n_};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(copy_policy_kernel<LDM, STM>), dim3(blocks), dim3(256), args, 0, s,
                              a, b, 0));
    };
    L.v.push_back(v);
}

template <int LEVEL>
static void add_plus(Lab &L, int wg_per_cu, int out_idx = 0)
{
    P p = L.pA;
    p.out = L.outs[out_idx];
    int blocks = L.cus * wg_per_cu;
    blocks -= blocks % 8;
    char name[64];
    snprintf(name, sizeof name, "copy_plus_level%d_wg%d_out%d", LEVEL, wg_per_cu, out_idx);
    Variant v;
    v.name = name; v.blocks = blocks; v.threads = 256; v.bytes = 2.0 * p.npix; v.check = LEVEL == 3 ? p.out : nullptr;
    hipStream_t s = L.s;
    v.launch = [p, blocks, s](hipEvent_t a, hipEvent_t b) {
        P pp = p;
        void *args[] = {&pp};
        CK(hipExtLaunchKernel(reinterpret_cast<const void *>(copy_plus_kernel<LEVEL>), dim3(blocks), dim3(256), args, 0, s, a, b, 0));
    };
    L.v.push_back(v);
}

static double median(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static void run_all(Lab &L, int rounds, int per_round)
{
    std::vector<hipEvent_t> e0(per_round), e1(per_round);
    for (int i = 0; i < per_round; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    // first launch of every variant: correctness
    for (auto &v : L.v) {
        if (v.check)
            CK(hipMemsetAsync(v.check, 0x5a, L.pA.npix, L.s));
        v.launch(nullptr, nullptr);
        if (v.check) {
            CK(hipMemsetAsync(L.d_count, 0, 8, L.s));
            hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, L.s, (const u32x4 *)v.check, (const u32x4 *)L.ref,
                               (size_t)L.pA.npix / 16, L.d_count);
            unsigned long long c;
            CK(hipMemcpyAsync(&c, L.d_count, 8, hipMemcpyDeviceToHost, L.s));
            CK(hipStreamSynchronize(L.s));
            v.bad = (long long)c;
        }
    }
    CK(hipStreamSynchronize(L.s));
    for (int r = 0; r < rounds; r++) {
        for (auto &v : L.v) {
            v.launch(nullptr, nullptr);                 // one untimed launch after the switch of kernels
            for (int i = 0; i < per_round; i++)
                v.launch(e0[i], e1[i]);
            CK(hipStreamSynchronize(L.s));
            for (int i = 0; i < per_round; i++) {
                float ms;
                CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
                v.ms.push_back(ms);
            }
        }
    }
    for (auto &v : L.v) {
        double sum = 0;
        for (float m : v.ms) sum += m;
        const double avg = sum / v.ms.size(), med = median(v.ms), mn = *std::min_element(v.ms.begin(), v.ms.end());
        printf("{\"variant\": \"%s\", \"blocks\": %d, \"threads\": %d, \"n\": %zu, \"avg_ms\": %.4f, \"median_ms\": %.4f, "
               "\"min_ms\": %.4f, \"GBps_median\": %.1f, \"frac_median\": %.4f, \"bad_vectors\": %lld, \"round_avgs\": [",
               v.name.c_str(), v.blocks, v.threads, v.ms.size(), avg, med, mn, v.bytes / med / 1e6, v.bytes / med / 1e6 / 8000.0,
               v.bad);
        for (int r = 0; r < rounds; r++) {
            double rs = 0;
            for (int i = 0; i < per_round; i++) rs += v.ms[r * per_round + i];
            printf("%s%.4f", r ? ", " : "", rs / per_round);
        }
        printf("]}\n");
    }
    fflush(stdout);
}

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// LAB_SPREAD=1: the three output rasters are virtual ranges backed by 32 MiB physical chunks created with
// 256 MiB of ballast between them (what gcn10_gpu_malloc_spread does): written ~10 % faster than a plain one.
static uint8_t *spread_alloc(size_t bytes)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)32 << 20, n = (bytes + chunk - 1) / chunk;
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, n * chunk, 0, nullptr, 0));
    std::vector<void *> ballast;
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(va + i * chunk, chunk, 0, h, 0));
        void *b = nullptr;
        if (hipMalloc(&b, (size_t)256 << 20) == hipSuccess)
            ballast.push_back(b);
        else
            (void)hipGetLastError();
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    for (void *b : ballast)
        CK(hipFree(b));
    return (uint8_t *)va;
}

int main(int argc, char **argv)
{
    const uint32_t W = argc > 1 ? atoi(argv[1]) : 36000, rows = argc > 2 ? atoi(argv[2]) : 36000;
    Lab L;
    const int rounds = argc > 3 ? atoi(argv[3]) : 6;
    const int per_round = argc > 4 ? atoi(argv[4]) : 6;
    const char *set = argc > 5 ? argv[5] : "main";
    const uint32_t npix = W * rows, hs = 1440;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    L.cus = prop.multiProcessorCount;
    CK(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));

    // ---- synthetic block: classes and probabilities of bench.py's "iid" pattern ----
    static const uint8_t classes[12] = {0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100};
    std::vector<uint8_t> esa((size_t)npix);
    uint32_t seed = 1;
    for (size_t i = 0; i < npix; i++) {
        uint32_t r = lcg(seed) % 100;
        esa[i] = r < 30 ? 0 : classes[1 + (r - 30) % 11];
    }
    static const uint8_t soils[10] = {0, 1, 2, 3, 4, 11, 12, 13, 14, 255};
    std::vector<uint8_t> coarse((size_t)hs * hs);
    for (auto &c : coarse) c = soils[lcg(seed) % 10];
    std::vector<int32_t> ci(W), cj(rows);
    for (uint32_t x = 0; x < W; x++) ci[x] = std::min<uint32_t>((uint32_t)(((double)x + 0.5) * hs / W + 0.5), hs - 1);
    for (uint32_t y = 0; y < rows; y++) cj[y] = std::min<uint32_t>((uint32_t)(((double)y + 0.5) * hs / rows + 0.5), hs - 1);
    const uint32_t stride = ((W + 15u) & ~15u) + 32u;       // 16 B in front, >= 16 B behind
    std::vector<uint8_t> hxA((size_t)stride * hs + 32, 0x55), hxB((size_t)stride * hs + 32, 0x81);
    for (uint32_t r = 0; r < hs; r++)
        for (uint32_t x = 0; x < W; x++) {
            uint8_t h = coarse[(size_t)r * hs + ci[x]];
            uint8_t d = (h >= 11 && h <= 14) ? 4 : (h < 5 ? h : 5);       // drained plane
            hxA[16 + (size_t)r * stride + x] = d | (d << 4);
            hxB[16 + (size_t)r * stride + x] = (d & 3) | ((d >> 2) << 7);
        }
    // table: arbitrary but full (every class has a value) so that any wrong index shows
    std::vector<uint8_t> lutA((kLutA + 15) & ~15, 255), lutB(kLutB, 255);
    for (int s = 0; s < 6; s++)
        for (int lc = 0; lc < 256; lc++) {
            uint8_t v = s == 5 ? 255 : (uint8_t)((lc * 7 + s * 31 + 3) % 251);
            lutA[s * kPlane1 + lc] = v;
            for (int rep = 0; rep < 32; rep++)
                lutB[(lc << 8) | (s & 3) | (rep << 2) | ((s >> 2) << 7)] = v;
        }

    uint8_t *d_esa, *d_hxA, *d_hxB, *d_lutA, *d_lutB, *d_ref;
    int32_t *d_cj;
    CK(hipMalloc((void **)&d_esa, npix)); CK(hipMalloc((void **)&d_ref, npix));
    const bool spread = getenv("LAB_SPREAD") && atoi(getenv("LAB_SPREAD"));
    for (int i = 0; i < 3; i++) {
        if (spread)
            L.outs[i] = spread_alloc(npix);
        else
            CK(hipMalloc((void **)&L.outs[i], npix));
    }
    CK(hipMalloc((void **)&d_hxA, hxA.size())); CK(hipMalloc((void **)&d_hxB, hxB.size()));
    CK(hipMalloc((void **)&d_lutA, lutA.size())); CK(hipMalloc((void **)&d_lutB, lutB.size()));
    CK(hipMalloc((void **)&d_cj, rows * 4)); CK(hipMalloc((void **)&L.d_count, 8));
    CK(hipMemcpy(d_esa, esa.data(), npix, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_hxA, hxA.data(), hxA.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_hxB, hxB.data(), hxB.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lutA, lutA.data(), lutA.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lutB, lutB.data(), lutB.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cj, cj.data(), rows * 4, hipMemcpyHostToDevice));
    L.ref = d_ref;
    P base{};
    base.esa = d_esa; base.cj = d_cj; base.out = L.outs[0]; base.W = W; base.rows = rows; base.npix = npix; base.nvec = npix / 16;
    base.hx_stride = stride; base.hx_rows = hs;
    L.pA = base; L.pA.hx = d_hxA + 16; L.pA.lut = d_lutA;
    L.pB = base; L.pB.hx = d_hxB + 16; L.pB.lut = d_lutB;
    hipLaunchKernelGGL(ref_kernel, dim3(4096), dim3(256), 0, L.s, d_esa, d_hxA + 16, stride, d_cj, d_lutA, d_ref, W, npix);
    CK(hipStreamSynchronize(L.s));
    printf("{\"lab\": \"strip\", \"W\": %u, \"rows\": %u, \"cus\": %d, \"rounds\": %d, \"per_round\": %d, \"set\": \"%s\", "
           "\"out_ptrs\": [\"%p\", \"%p\", \"%p\"], \"esa_ptr\": \"%p\"}\n",
           W, rows, L.cus, rounds, per_round, set, (void *)L.outs[0], (void *)L.outs[1], (void *)L.outs[2], (void *)d_esa);

    if (!strcmp(set, "main")) {
        add_copy<2, 256, 1>(L, 8); add_copy<2, 256, 0>(L, 8); add_copy<2, 512, 1>(L, 4); add_copy<2, 256, 1>(L, 16);
        add_copy<4, 256, 0>(L, 16);
        add_variant<0, 2, true, 256, 0>(L, 8);              // run 1's best
        add_variant<0, 2, true, 256, 0>(L, 16);
        add_variant<0, 2, true, 256, 0, 0>(L, 8);           // grid-stride instead of XCD slabs
        add_variant<0, 2, true, 256, 0, 0>(L, 16);
        add_variant<0, 2, false, 256, 0>(L, 8);
        add_variant<0, 4, false, 256, 0, 0>(L, 16);
        add_variant<0, 4, false, 256, 0, 1>(L, 8);
        add_variant<0, 2, true, 256, 0, 1, 2>(L, 8);        // landcover loads with the default policy
        add_variant<0, 2, true, 256, 0, 1, 1>(L, 8);        // raster stores with the default policy
        add_variant<0, 2, true, 256, 0, 1, 3>(L, 8, 1);     // the same kernel into two other allocations
        add_variant<0, 2, true, 256, 0, 1, 3>(L, 8, 2);
        add_variant<0, 2, true, 256, 1>(L, 8);              // timing only: no gather / no soil load / neither
        add_variant<0, 2, true, 256, 2>(L, 8);
        add_variant<0, 2, true, 256, 3>(L, 8);
    }
    else if (!strcmp(set, "gap")) {
        // where is the strip kernel's time against the copy, in well-placed memory?
        add_copy<2, 256, 1>(L, 8); add_copy<2, 256, 1>(L, 16);
        add_plus<1>(L, 8, 0); add_plus<2>(L, 8, 0); add_plus<3>(L, 8, 0); add_plus<4>(L, 8, 0); add_plus<5>(L, 8, 0);
        add_variant<0, 2, true, 256, 0>(L, 8); add_variant<0, 2, true, 256, 0>(L, 16);
        add_variant<0, 4, false, 256, 0>(L, 16); add_variant<0, 2, false, 256, 0>(L, 8);
        add_variant<0, 1, true, 256, 0>(L, 16); add_variant<0, 4, true, 256, 0>(L, 8);
        add_variant<0, 2, true, 256, 4>(L, 8);              // timing only: compact soil words (4 B per 16 px)
        add_variant<0, 4, false, 256, 4>(L, 16);
        add_variant<0, 4, true, 256, 4>(L, 8);
        add_variant<0, 2, true, 256, 4>(L, 16);
        add_variant<0, 2, true, 256, 1>(L, 8);              // timing only: no gather / no soil load / neither
        add_variant<0, 2, true, 256, 2>(L, 8);
        add_variant<0, 2, true, 256, 3>(L, 8);
        add_variant<0, 4, false, 256, 1>(L, 16); add_variant<0, 4, false, 256, 2>(L, 16); add_variant<0, 4, false, 256, 3>(L, 16);
    }
    else if (!strcmp(set, "plus")) {
        for (int o = 0; o < 2; o++) {
            add_plus<0>(L, 8, o); add_plus<1>(L, 8, o); add_plus<2>(L, 8, o); add_plus<3>(L, 8, o);
            add_variant<0, 2, false, 256, 0, 1>(L, 8, o);
            add_variant<0, 2, true, 256, 0, 1>(L, 8, o);
        }
        add_copy<2, 256, 1>(L, 8);
    }
    else if (!strcmp(set, "regions")) {
        // the same copies / kernels into three allocations: slabs vs grid-stride vs region-synchronous sub-slabs
        add_copy<2, 256, 1>(L, 8); add_copy<2, 256, 0>(L, 8); add_copy<2, 256, 2>(L, 8); add_copy<2, 256, 2>(L, 16);
        add_copy<4, 256, 2>(L, 8);
        for (int o = 0; o < 3; o++) {
            add_variant<0, 2, true, 256, 0, 1>(L, 8, o);
            add_variant<0, 2, false, 256, 0, 2>(L, 8, o);
            add_variant<0, 4, false, 256, 0, 2>(L, 8, o);
            add_variant<0, 4, false, 256, 0, 1>(L, 16, o);
        }
    }
    else if (!strcmp(set, "policy")) {
        add_copy<2, 256, 1>(L, 8);
        add_policy<0, 1>(L, 8); add_policy<0, 2>(L, 8); add_policy<0, 3>(L, 8); add_policy<0, 4>(L, 8);
        add_policy<1, 1>(L, 8); add_policy<1, 2>(L, 8); add_policy<2, 2>(L, 8); add_policy<2, 3>(L, 8);
        add_copy<2, 256, 1>(L, 8);
    }
    else if (!strcmp(set, "phased")) {
        add_copy<2, 256, 1>(L, 8); add_copy<4, 256, 1>(L, 8); add_copy<2, 256, 1>(L, 4);
        add_phased<2, 256, 6>(L, 8); add_phased<2, 256, 7>(L, 8); add_phased<2, 256, 8>(L, 8); add_phased<2, 256, 9>(L, 8);
        add_phased<4, 256, 7>(L, 8); add_phased<4, 256, 8>(L, 8); add_phased<4, 256, 9>(L, 8); add_phased<4, 256, 10>(L, 8);
        add_phased<8, 256, 8>(L, 4); add_phased<8, 256, 9>(L, 4); add_phased<8, 256, 10>(L, 4);
    }
    else {      // "check": correctness of the odd-width paths
        add_variant<0, 2, true, 256, 0>(L, 8);
        add_variant<0, 1, true, 256, 0>(L, 8);
        add_variant<0, 2, false, 256, 0, 0>(L, 8);
        add_variant<0, 4, false, 256, 0>(L, 8);
        add_variant<2, 2, true, 1024, 0>(L, 2);
    }
    run_all(L, rounds, per_round);
    return 0;
}

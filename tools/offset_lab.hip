// offset_lab.hip -- how much does the distance between the read stream and the write stream of a
// 1R:1W streaming kernel matter on this device?  One arena; the source sits at its start, the
// destination at arena + D for a list of D; every (D, map) pair is timed in interleaved rounds with
// events attached to the dispatch.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/offset_lab tools/offset_lab.hip
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

template <int UN, int MAP>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *in, u32x4 *out, size_t nvec)
{
    const size_t nchunk = (nvec + 256 * UN - 1) / (256 * UN);
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = MAP ? xcd * per + (blockIdx.x >> 3) : blockIdx.x;
    const size_t cend = MAP ? ((xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk) : nchunk;
    const size_t cstep = MAP ? gridDim.x / 8 : gridDim.x;
    for (; c < cend; c += cstep) {
        u32x4 v[UN];
        size_t idx[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            idx[u] = (c * UN + u) * 256 + threadIdx.x;
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0));
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
    }
}

struct Case { size_t D; int map; std::vector<float> ms; };

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)36000 * 36000, nvec = bytes / 16;
    const size_t base = 0x4D600000;                 // 1.296 GB rounded up to 2 MiB: hipMalloc's natural stride
    const int rounds = argc > 1 ? atoi(argv[1]) : 4, per = 5;
    uint8_t *arena;
    const size_t arena_bytes = (size_t)(argc > 6 ? atoi(argv[6]) : 7) << 30;
    CK(hipMalloc((void **)&arena, arena_bytes));
    CK(hipMemset(arena, 1, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<size_t> Ds;
    const char *mode = argc > 2 ? argv[2] : "sweep";
    if (!strcmp(mode, "list")) {
        // explicit distances in MiB: argv[3] = comma separated list
        const char *q = argc > 3 ? argv[3] : "1536";
        while (*q) {
            Ds.push_back((size_t)strtod(q, (char **)&q) * 1048576.0);
            if (*q == ',') q++;
        }
    }
    else if (!strcmp(mode, "bits")) {
        // which address bits of D matter: single-bit flips of an ordinary D and of the one fast D of run 1
        const size_t fast = argc > 3 ? strtoull(argv[3], nullptr, 0) : 0x1492a6000ull;
        Ds.push_back(base);
        Ds.push_back(fast);
        for (int b = 8; b <= 32; b++) {
            Ds.push_back(base ^ ((size_t)1 << b));
            Ds.push_back(fast ^ ((size_t)1 << b));
        }
    }
    else {
        const size_t deltas[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20,
                                 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20};
        for (size_t d : deltas) Ds.push_back(base + d);
        for (int k = 2; k <= 4; k++) Ds.push_back(k * base);
        uint32_t seed = argc > 3 ? atoi(argv[3]) : 12345;
        const int nrand = argc > 4 ? atoi(argv[4]) : 10;
        for (int i = 0; i < nrand; i++) {               // multiples of 4 KiB at random
            seed = seed * 1664525u + 1013904223u;
            Ds.push_back(base + ((size_t)(seed >> 10) % (1u << 20)) * 4096);
        }
    }
    const int only_map = argc > 5 ? atoi(argv[5]) : -1;
    std::vector<Case> cases;
    for (size_t D : Ds)
        for (int map = 0; map < 2; map++)
            if (D + bytes <= arena_bytes && D >= bytes && (only_map < 0 || map == only_map))
                cases.push_back({D, map, {}});
    hipEvent_t e0[8], e1[8];
    for (int i = 0; i < per; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    printf("{\"lab\": \"offset\", \"arena\": \"%p\", \"bytes\": %zu, \"cases\": %zu}\n", (void *)arena, bytes, cases.size());
    for (int r = 0; r < rounds; r++)
        for (auto &c : cases) {
            const u32x4 *in = (const u32x4 *)arena;
            u32x4 *out = (u32x4 *)(arena + c.D);
            size_t n = nvec;
            void *args[] = {&in, &out, &n};
            const void *fn = c.map ? (const void *)copy_kernel<2, 1> : (const void *)copy_kernel<2, 0>;
            CK(hipExtLaunchKernel(fn, dim3(2048), dim3(256), args, 0, s, nullptr, nullptr, 0));
            for (int i = 0; i < per; i++)
                CK(hipExtLaunchKernel(fn, dim3(2048), dim3(256), args, 0, s, e0[i], e1[i], 0));
            CK(hipStreamSynchronize(s));
            for (int i = 0; i < per; i++) { float ms; CK(hipEventElapsedTime(&ms, e0[i], e1[i])); c.ms.push_back(ms); }
        }
    for (auto &c : cases) {
        std::sort(c.ms.begin(), c.ms.end());
        const double med = c.ms[c.ms.size() / 2];
        printf("{\"D\": %zu, \"D_hex\": \"0x%zx\", \"delta\": %zd, \"map\": %d, \"median_ms\": %.4f, \"min_ms\": %.4f, \"GBps\": %.1f}\n",
               c.D, c.D, (ssize_t)(c.D - base), c.map, med, c.ms[0], 2.0 * bytes / med / 1e6);
    }
    return 0;
}

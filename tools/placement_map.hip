// placement_map.hip -- does the speed of a 1R:1W stream depend on WHERE in a large allocation the
// source and the destination lie?  One arena of A GiB; copies of C MiB from every source position to
// every destination position on a G-GiB grid, plus read-only and write-only passes per position.
// Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/placement_map tools/placement_map.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// MODE 0 copy, 1 read only (sum kept from being optimised away), 2 write only
template <int MODE>
__global__ __launch_bounds__(256) void k(const u32x4 *in, u32x4 *out, size_t nvec, uint32_t *sink)
{
    const size_t nchunk = (nvec + 511) / 512;
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = xcd * per + (blockIdx.x >> 3);
    const size_t cend = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk, cstep = gridDim.x / 8;
    u32x4 acc = {0, 0, 0, 0};
    for (; c < cend; c += cstep) {
        u32x4 v[2];
        size_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (c * 2 + u) * 256 + threadIdx.x;
            if (MODE != 2) v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0));
            else v[u] = u32x4{(uint32_t)idx[u], 1u, 2u, 3u};
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (MODE == 1) acc ^= v[u];
            else if (idx[u] < nvec) __builtin_nontemporal_store(v[u], out + idx[u]);
        }
    }
    if (MODE == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

int main(int argc, char **argv)
{
    const size_t GiB = (size_t)1 << 30;
    const size_t A = (argc > 1 ? atoi(argv[1]) : 40) * GiB;
    const size_t C = (size_t)(argc > 2 ? atoi(argv[2]) : 512) << 20;
    const size_t G = (size_t)(argc > 3 ? atoi(argv[3]) : 2048) << 20;     // grid step in MiB
    uint8_t *arena;
    uint32_t *sink;
    CK(hipMalloc((void **)&arena, A));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(arena, 1, A));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0[4], e1[4];
    for (int i = 0; i < 4; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    std::vector<size_t> pos;
    for (size_t p = 0; p + C <= A; p += G) pos.push_back(p);
    const size_t nvec = C / 16;
    printf("{\"lab\": \"placement\", \"arena\": \"%p\", \"arena_GiB\": %zu, \"copy_MiB\": %zu, \"grid_MiB\": %zu, \"positions\": %zu}\n",
           (void *)arena, A / GiB, C >> 20, G >> 20, pos.size());
    auto run = [&](int mode, size_t ps, size_t pd) {
        const u32x4 *in = (const u32x4 *)(arena + ps);
        u32x4 *out = (u32x4 *)(arena + pd);
        size_t n = nvec;
        void *args[] = {&in, &out, &n, &sink};
        const void *fn = mode == 0 ? (const void *)k<0> : mode == 1 ? (const void *)k<1> : (const void *)k<2>;
        CK(hipExtLaunchKernel(fn, dim3(2048), dim3(256), args, 0, s, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++)
            CK(hipExtLaunchKernel(fn, dim3(2048), dim3(256), args, 0, s, e0[i], e1[i], 0));
        CK(hipStreamSynchronize(s));
        float best = 1e9f, ms;
        std::vector<float> v;
        for (int i = 0; i < 3; i++) { CK(hipEventElapsedTime(&ms, e0[i], e1[i])); v.push_back(ms); }
        std::sort(v.begin(), v.end());
        (void)best;
        return v[1];
    };
    for (int mode = 1; mode <= 2; mode++) {
        printf("{\"mode\": \"%s\", \"GBps\": [", mode == 1 ? "read" : "write");
        for (size_t i = 0; i < pos.size(); i++)
            printf("%s%.0f", i ? ", " : "", C / run(mode, pos[i], pos[i]) / 1e6);
        printf("]}\n");
        fflush(stdout);
    }
    for (size_t i = 0; i < pos.size(); i++) {
        printf("{\"mode\": \"copy\", \"src_GiB\": %.2f, \"GBps_by_dst\": [", (double)pos[i] / GiB);
        for (size_t j = 0; j < pos.size(); j++) {
            const bool overlap = (pos[i] < pos[j] + C) && (pos[j] < pos[i] + C);
            printf("%s%.0f", j ? ", " : "", overlap ? 0.0 : 2.0 * C / run(0, pos[i], pos[j]) / 1e6);
        }
        printf("]}\n");
        fflush(stdout);
    }
    return 0;
}

// region_lab: does the ABSOLUTE place of a buffer in VRAM decide the copy rate?  (GPU only; lab.)
// Allocates N physically contiguous 1.296 GB buffers one after another, then times: a read-only sweep and a
// write-only sweep of every buffer, and the slab copy from four chosen sources into every buffer.
//   hipcc -O3 --offload-arch=gfx950 -o tools/region_lab tools/region_lab.hip && tools/region_lab [N=40] [contiguous=1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

// mode 0 copy, 1 read only (result folded into a never-true store), 2 write only
template <int MODE>
__global__ __launch_bounds__(kThreads) void sweep(const u32x4 *in, u32x4 *out, uint32_t nvec, uint32_t ntrips)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t per = (ntrips + 7u) / 8u;
    const uint32_t lo = (b & 7u) * per;
    const uint32_t hi = lo + per < ntrips ? lo + per : ntrips;
    u32x4 acc = { 0, 0, 0, 0 };
    for (uint32_t trip = lo + (b >> 3); trip < hi; trip += nb / 8u) {
        u32x4 v[2];
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (trip * 2u + u) * (uint32_t)kThreads + threadIdx.x;
            if (MODE != 2)
                v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0u));
            else
                v[u] = u32x4{ trip, trip, trip, trip };
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (MODE == 1)
                acc ^= v[u];
            else if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
        }
    }
    if (MODE == 1 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
        out[0] = acc;
}

static hipEvent_t e0[5], e1[5];

template <int MODE>
static float time_it(const void *src, void *dst, size_t bytes)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    void *args[] = { &in, &out, &nvec, &ntrips };
    hipLaunchKernelGGL(sweep<MODE>, dim3(2048), dim3(kThreads), 0, 0, in, out, nvec, ntrips);
    for (int k = 0; k < 5; k++)
        CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(sweep<MODE>), dim3(2048), dim3(kThreads), args, 0, 0,
                                 e0[k], e1[k], 0));
    CHECK(hipDeviceSynchronize());
    float ms[5];
    for (int k = 0; k < 5; k++)
        CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
    std::sort(ms, ms + 5);
    return ms[2];
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 40;
    bool contiguous = !(argc > 2 && atoi(argv[2]) == 0);
    size_t bytes = 1296000000ull;
    for (int k = 0; k < 5; k++) {
        CHECK(hipEventCreate(&e0[k]));
        CHECK(hipEventCreate(&e1[k]));
    }
    std::vector<char *> buf(n);
    for (int i = 0; i < n; i++) {
        if (contiguous)
            CHECK(hipExtMallocWithFlags((void **)&buf[i], bytes, hipDeviceMallocContiguous));
        else
            CHECK(hipMalloc((void **)&buf[i], bytes));
        CHECK(hipMemset(buf[i], i + 1, bytes));
    }
    const char *mem = contiguous ? "contiguous" : "hipMalloc";
    printf("{\"memory\": \"%s\", \"what\": \"virtual addresses\", \"ptr\": [", mem);
    for (int i = 0; i < n; i++)
        printf("%s\"%p\"", i ? ", " : "", (void *)buf[i]);
    printf("]}\n");
    printf("{\"memory\": \"%s\", \"what\": \"read-only sweep of buffer i\", \"ms\": [", mem);
    for (int i = 0; i < n; i++)
        printf("%s%.4f", i ? ", " : "", time_it<1>(buf[i], buf[i], bytes));
    printf("]}\n");
    printf("{\"memory\": \"%s\", \"what\": \"write-only sweep of buffer i\", \"ms\": [", mem);
    for (int i = 0; i < n; i++)
        printf("%s%.4f", i ? ", " : "", time_it<2>(buf[i], buf[i], bytes));
    printf("]}\n");
    fflush(stdout);
    const int srcs[] = { 0, 1, n / 2, n - 1 };
    for (int s : srcs) {
        printf("{\"memory\": \"%s\", \"what\": \"copy buffer %d -> buffer i\", \"ms\": [", mem, s);
        for (int i = 0; i < n; i++) {
            if (i == s) {
                printf("%snull", i ? ", " : "");
                continue;
            }
            printf("%s%.4f", i ? ", " : "", time_it<0>(buf[s], buf[i], bytes));
        }
        printf("]}\n");
        fflush(stdout);
    }
    return 0;
}

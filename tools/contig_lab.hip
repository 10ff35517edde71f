// contig_lab: does the PHYSICAL layout of an allocation explain the fast and slow rasters?  (GPU only; lab.)
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/contig_lab tools/contig_lab.hip
//   tools/contig_lab [bytes=1296000000] [buffers=6]
//
// Times the 1R:1W slab copy (the same shape as gcn10_gpu_stream_copy) from one source into several
// destinations, for three ways of obtaining the memory:
//   default     hipMalloc per buffer
//   contiguous  hipExtMallocWithFlags(hipDeviceMallocContiguous) per buffer (physically contiguous VRAM)
//   arena       one contiguous arena; source at 0, destinations at increasing offsets (so the physical distance
//               between source and destination is known)
// Prints one JSON line per mode.
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

__global__ __launch_bounds__(kThreads) void slab_copy(const u32x4 *in, u32x4 *out, uint32_t nvec, uint32_t ntrips)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t per = (ntrips + 7u) / 8u;
    const uint32_t lo = (b & 7u) * per;
    const uint32_t hi = lo + per < ntrips ? lo + per : ntrips;
    for (uint32_t trip = lo + (b >> 3); trip < hi; trip += nb / 8u) {
        u32x4 v[2];
        uint32_t idx[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            idx[u] = (trip * 2u + u) * (uint32_t)kThreads + threadIdx.x;
            v[u] = __builtin_nontemporal_load(in + (idx[u] < nvec ? idx[u] : 0u));
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (idx[u] < nvec)
                __builtin_nontemporal_store(v[u], out + idx[u]);
    }
}

static hipEvent_t e0[5], e1[5];

static float time_copy(const void *src, void *dst, size_t bytes)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    uint32_t grid = 256 * 8;
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    void *args[] = { &in, &out, &nvec, &ntrips };
    hipLaunchKernelGGL(slab_copy, dim3(grid), dim3(kThreads), 0, 0, in, out, nvec, ntrips);
    for (int k = 0; k < 5; k++)
        CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(slab_copy), dim3(grid), dim3(kThreads), args, 0, 0,
                                 e0[k], e1[k], 0));
    CHECK(hipDeviceSynchronize());
    float ms[5];
    for (int k = 0; k < 5; k++)
        CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
    std::sort(ms, ms + 5);
    return ms[2];
}

static void *get(size_t bytes, bool contiguous)
{
    void *p = nullptr;
    if (contiguous) {
        hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            fprintf(stderr, "contiguous allocation of %zu bytes refused: %s\n", bytes, hipGetErrorString(e));
            (void)hipGetLastError();
            return nullptr;
        }
    }
    else
        CHECK(hipMalloc(&p, bytes));
    return p;
}

// A raster whose virtual range is backed by separately created physical chunks, mapped in a chosen order.
struct Scattered {
    char *va = nullptr;
    size_t size = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
};

static bool scattered_alloc(Scattered &s, size_t bytes, size_t chunk, int order, unsigned seed)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) {
        fprintf(stderr, "no virtual memory management on this device\n");
        return false;
    }
    chunk = (chunk + gran - 1) / gran * gran;
    size_t n = (bytes + chunk - 1) / chunk;
    s.size = n * chunk;
    s.chunk = chunk;
    CHECK(hipMemAddressReserve((void **)&s.va, s.size, 0, nullptr, 0));
    s.h.resize(n);
    for (size_t i = 0; i < n; i++)
        CHECK(hipMemCreate(&s.h[i], chunk, &prop, 0));
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; i++)
        perm[i] = i;
    if (order == 1) {           // shuffle
        uint64_t x = 0x9e3779b97f4a7c15ull * (seed + 1);
        for (size_t i = n - 1; i > 0; i--) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(perm[i], perm[x % (i + 1)]);
        }
    }
    else if (order == 2)        // reversed
        std::reverse(perm.begin(), perm.end());
    else if (order == 3) {      // stride: 0, n/8, 2n/8 ... then 1, n/8+1 ...
        size_t k = 0, cols = (n + 7) / 8;
        for (size_t c = 0; c < cols; c++)
            for (size_t r = 0; r < 8; r++)
                if (r * cols + c < n)
                    perm[k++] = r * cols + c;
    }
    for (size_t i = 0; i < n; i++)
        CHECK(hipMemMap(s.va + i * chunk, chunk, 0, s.h[perm[i]], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(s.va, s.size, &acc, 1));
    return true;
}

// (the lab never unmaps: an unmap / release / re-reserve cycle aborted inside the runtime on ROCm 7.2, and the
// process is short-lived; 24 configurations x 3 rasters stay far below the card's memory)
static void scattered_free(Scattered &s)
{
    s = Scattered();
}

int main(int argc, char **argv)
{
    // tools/contig_lab [rasters per recipe = 8] [source recipe 0..5 = 0]
    // Allocates the rasters of all recipes round-robin (so no recipe owns a phase of the process), then times the
    // copy from one source into each, three rounds, and prints the best time of every raster per recipe.
    size_t bytes = 1296000000ull;
    int per = argc > 1 ? atoi(argv[1]) : 8;
    int src_recipe = argc > 2 ? atoi(argv[2]) : 0;
    for (int k = 0; k < 5; k++) {
        CHECK(hipEventCreate(&e0[k]));
        CHECK(hipEventCreate(&e1[k]));
    }
    static const char *names[] = { "hipMalloc", "contiguous flag", "2 MiB chunks in creation order",
                                   "2 MiB chunks reversed", "2 MiB chunks shuffled", "32 MiB chunks in creation order" };
    const int nrec = 6;
    auto make = [&](int recipe, unsigned seed) -> void * {
        if (recipe == 0)
            return get(bytes, false);
        if (recipe == 1)
            return get(bytes, true);
        Scattered sc;
        size_t chunk = recipe == 5 ? (32u << 20) : (2u << 20);
        int order = recipe == 3 ? 2 : recipe == 4 ? 1 : 0;
        if (!scattered_alloc(sc, bytes, chunk, order, seed))
            exit(3);
        return sc.va;
    };
    void *src = make(src_recipe, 1000);
    CHECK(hipMemset(src, 0x5a, bytes));
    std::vector<std::vector<void *>> dst(nrec);
    for (int i = 0; i < per; i++)
        for (int r = 0; r < nrec; r++)
            dst[r].push_back(make(r, (unsigned)i));
    std::vector<std::vector<float>> best(nrec, std::vector<float>(per, 1e30f));
    for (int rnd = 0; rnd < 3; rnd++)
        for (int i = 0; i < per; i++)
            for (int r = 0; r < nrec; r++)
                best[r][i] = std::min(best[r][i], time_copy(src, dst[r][i], bytes));
    for (int r = 0; r < nrec; r++) {
        printf("{\"source\": \"%s\", \"destination\": \"%s\", \"copy_ms\": [", names[src_recipe], names[r]);
        for (int i = 0; i < per; i++)
            printf("%s%.4f", i ? ", " : "", best[r][i]);
        printf("]}\n");
    }
    return 0;
}

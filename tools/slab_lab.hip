// slab_lab: in PHYSICALLY CONTIGUOUS memory the sixteen streams of a slab copy (8 XCDs x read + write) keep a
// constant distance from each other for the whole launch.  Does choosing those distances (mod 128 MiB) move the
// copy from the 0.45 ms level to the 0.40 ms level seen with lucky scattered allocations?  (GPU only; lab.)
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/slab_lab tools/slab_lab.hip && tools/slab_lab
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
constexpr int kMaxPass = 4;
constexpr uint32_t kTripBytes = 2 * kThreads * 16;     // 8 KiB

struct Plan {
    int npass;
    uint32_t lo[8][kMaxPass], hi[8][kMaxPass];          // in trips
};

__global__ __launch_bounds__(kThreads) void plan_copy(const u32x4 *in, u32x4 *out, uint32_t nvec, const Plan plan)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7u;
    for (int p = 0; p < plan.npass; p++) {
        const uint32_t hi = plan.hi[x][p];
        for (uint32_t trip = plan.lo[x][p] + (b >> 3); trip < hi; trip += nb / 8u) {
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
}

// One pass over equal slabs, but the 2 MiB steps of a slab are visited in a permuted order: step k of XCD x is
// chunk (k * mult + x * skew) mod K, walked backwards on odd XCDs if reverse_odd.  (Same bytes moved, same
// rasters; only WHEN each 2 MiB piece is touched changes.)
__global__ __launch_bounds__(kThreads) void order_copy(const u32x4 *in, u32x4 *out, uint32_t nvec, uint32_t ntrips,
                                                       uint32_t mult, uint32_t skew, uint32_t reverse_odd)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7u;
    const uint32_t per = (ntrips + 7u) / 8u;
    const uint32_t lo = x * per;
    const uint32_t hi = lo + per < ntrips ? lo + per : ntrips;
    const uint32_t w = nb / 8u;                         // trips per step
    const uint32_t K = (per + w - 1u) / w;
    for (uint32_t k = 0; k < K; k++) {
        uint32_t c = (k * mult + x * skew) % K;
        if (reverse_odd && (x & 1u))
            c = K - 1u - c;
        const uint32_t trip = lo + c * w + (b >> 3);
        if (trip >= hi)
            continue;
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

// Every workgroup streams its OWN contiguous band of the buffer (band = ntrips / gridDim trips), trip by trip:
// 2048 or 4096 sequential streams instead of 8 windows that all workgroups of an XCD share.  (What a kernel
// that gives each workgroup the raster rows of one soil row would do to the memory system.)
__global__ __launch_bounds__(kThreads) void band_copy(const u32x4 *in, u32x4 *out, uint32_t nvec, uint32_t ntrips,
                                                      uint32_t xcd_major)
{
    const uint32_t nb = gridDim.x;
    // xcd_major: consecutive bands go to the workgroups of one XCD (b & 7 = XCD), else round-robin over XCDs
    const uint32_t b = xcd_major ? (blockIdx.x & 7u) * (nb / 8u) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t per = (ntrips + nb - 1u) / nb;
    const uint32_t lo = b * per;
    const uint32_t hi = lo + per < ntrips ? lo + per : ntrips;
    for (uint32_t trip = lo; trip < hi; trip++) {
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

static float time_band(const void *src, void *dst, size_t bytes, uint32_t xcd_major, int bpc)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    uint32_t grid = 256 * bpc;
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    void *args[] = { &in, &out, &nvec, &ntrips, &xcd_major };
    hipLaunchKernelGGL(band_copy, dim3(grid), dim3(kThreads), 0, 0, in, out, nvec, ntrips, xcd_major);
    for (int k = 0; k < 5; k++)
        CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(band_copy), dim3(grid), dim3(kThreads), args, 0, 0,
                                 e0[k], e1[k], 0));
    CHECK(hipDeviceSynchronize());
    float ms[5];
    for (int k = 0; k < 5; k++)
        CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
    std::sort(ms, ms + 5);
    return ms[2];
}

static float time_order(const void *src, void *dst, size_t bytes, uint32_t mult, uint32_t skew, uint32_t rev, int bpc)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    uint32_t grid = 256 * bpc;
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    void *args[] = { &in, &out, &nvec, &ntrips, &mult, &skew, &rev };
    hipLaunchKernelGGL(order_copy, dim3(grid), dim3(kThreads), 0, 0, in, out, nvec, ntrips, mult, skew, rev);
    for (int k = 0; k < 5; k++)
        CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(order_copy), dim3(grid), dim3(kThreads), args, 0, 0,
                                 e0[k], e1[k], 0));
    CHECK(hipDeviceSynchronize());
    float ms[5];
    for (int k = 0; k < 5; k++)
        CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
    std::sort(ms, ms + 5);
    return ms[2];
}

static float time_copy(const void *src, void *dst, size_t bytes, const Plan &plan, int bpc)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t grid = 256 * bpc;
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    Plan pl = plan;
    void *args[] = { &in, &out, &nvec, &pl };
    hipLaunchKernelGGL(plan_copy, dim3(grid), dim3(kThreads), 0, 0, in, out, nvec, pl);
    for (int k = 0; k < 5; k++)
        CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(plan_copy), dim3(grid), dim3(kThreads), args, 0, 0,
                                 e0[k], e1[k], 0));
    CHECK(hipDeviceSynchronize());
    float ms[5];
    for (int k = 0; k < 5; k++)
        CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
    std::sort(ms, ms + 5);
    return ms[2];
}

// npass passes; pass p of XCD x covers trips [base_p + x*len_p, base_p + (x+1)*len_p), the last pass takes the rest
static Plan make_plan(uint32_t ntrips, int npass, const uint32_t *len_trips)
{
    Plan pl = {};
    pl.npass = npass;
    uint32_t base = 0;
    for (int p = 0; p < npass; p++) {
        uint32_t len = p + 1 < npass ? len_trips[p] : (ntrips - base + 7) / 8;
        for (int x = 0; x < 8; x++) {
            uint32_t lo = base + x * len, hi = lo + len;
            pl.lo[x][p] = lo < ntrips ? lo : ntrips;
            pl.hi[x][p] = hi < ntrips ? hi : ntrips;
        }
        base += 8 * len;
        if (base > ntrips)
            base = ntrips;
    }
    return pl;
}

int main(int argc, char **argv)
{
    size_t bytes = 1296000000ull;
    bool contiguous = !(argc > 1 && atoi(argv[1]) == 0);
    for (int k = 0; k < 5; k++) {
        CHECK(hipEventCreate(&e0[k]));
        CHECK(hipEventCreate(&e1[k]));
    }
    const size_t MiB = 1u << 20;
    const size_t first = 1280 * MiB;                    // a multiple of 128 MiB past the source
    const size_t arena_bytes = first + 160 * MiB + bytes;
    char *arena = nullptr;
    if (contiguous)
        CHECK(hipExtMallocWithFlags((void **)&arena, arena_bytes, hipDeviceMallocContiguous));
    else
        CHECK(hipMalloc((void **)&arena, arena_bytes));
    CHECK(hipMemset(arena, 0x5a, bytes));
    const uint32_t ntrips = (uint32_t)((bytes / 16 + 2 * kThreads - 1) / (2 * kThreads));
    const uint32_t t_per_mib = MiB / kTripBytes;

    struct Named { const char *name; Plan plan; };
    std::vector<Named> plans;
    plans.push_back({ "1 pass, equal slabs (154.5 MiB)", make_plan(ntrips, 1, nullptr) });
    {
        uint32_t l[] = { 80 * t_per_mib };
        plans.push_back({ "2 passes: 80 MiB slabs, then the rest (74.5)", make_plan(ntrips, 2, l) });
    }
    {
        uint32_t l[] = { 72 * t_per_mib };
        plans.push_back({ "2 passes: 72 MiB slabs, then the rest (82.5)", make_plan(ntrips, 2, l) });
    }
    {
        uint32_t l[] = { 40 * t_per_mib, 40 * t_per_mib, 40 * t_per_mib };
        plans.push_back({ "4 passes: 40 MiB slabs x3, then the rest (34.5)", make_plan(ntrips, 4, l) });
    }
    {
        uint32_t l[] = { 77 * t_per_mib + t_per_mib / 4 };
        plans.push_back({ "2 passes: equal halves (77.25 MiB slabs)", make_plan(ntrips, 2, l) });
    }
    {
        uint32_t l[] = { 48 * t_per_mib, 48 * t_per_mib };
        plans.push_back({ "3 passes: 48 MiB slabs x2, then the rest (58.5)", make_plan(ntrips, 3, l) });
    }
    for (int bpc = 8; bpc <= 8; bpc += 8)
        for (auto &np : plans) {
            printf("{\"memory\": \"%s\", \"blocks_per_cu\": %d, \"plan\": \"%s\", \"ms_by_D_minus_S_mod_16MiB\": [",
                   contiguous ? "contiguous" : "hipMalloc", bpc, np.name);
            for (int d = 0; d < 16; d++) {
                float best = 1e30f;
                for (int rnd = 0; rnd < 2; rnd++)
                    best = std::min(best, time_copy(arena, arena + first + (size_t)d * MiB, bytes, np.plan, bpc));
                printf("%s%.4f", d ? ", " : "", best);
            }
            printf("]}\n");
            fflush(stdout);
        }
    // one band per workgroup
    for (int bpc = 4; bpc <= 16; bpc *= 2)
        for (uint32_t xm = 0; xm < 2; xm++) {
            printf("{\"memory\": \"%s\", \"blocks_per_cu\": %d, \"plan\": \"one contiguous band per workgroup, %s\", "
                   "\"ms_by_D_minus_S_MiB\": [", contiguous ? "contiguous" : "hipMalloc", bpc,
                   xm ? "neighbouring bands on one XCD" : "bands dealt round-robin over the XCDs");
            for (int d = 0; d < 4; d++) {
                float best = 1e30f;
                for (int rnd = 0; rnd < 2; rnd++)
                    best = std::min(best, time_band(arena, arena + first + (size_t)d * MiB, bytes, xm, bpc));
                printf("%s%.4f", d ? ", " : "", best);
            }
            printf("]}\n");
            fflush(stdout);
        }
    // permuted visiting order (K = 78 steps of 2 MiB per XCD at 8 blocks per CU; 39 of 4 MiB at 16)
    {
        struct O { uint32_t mult, skew, rev; };
        const O orders[] = { {1, 0, 0}, {1, 0, 1}, {5, 0, 0}, {7, 0, 0}, {17, 0, 0}, {29, 0, 0}, {1, 5, 0}, {1, 10, 0},
                             {7, 3, 0}, {7, 3, 1}, {35, 0, 0}, {37, 11, 0} };
        for (int bpc = 8; bpc <= 16; bpc += 8)
            for (const O &o : orders) {
                printf("{\"memory\": \"%s\", \"blocks_per_cu\": %d, \"order\": {\"mult\": %u, \"skew\": %u, "
                       "\"reverse_odd\": %u}, \"ms_by_D_minus_S_MiB\": [", contiguous ? "contiguous" : "hipMalloc", bpc,
                       o.mult, o.skew, o.rev);
                for (int d = 0; d < 8; d++) {
                    float best = 1e30f;
                    for (int rnd = 0; rnd < 2; rnd++)
                        best = std::min(best, time_order(arena, arena + first + (size_t)d * MiB, bytes, o.mult, o.skew,
                                                         o.rev, bpc));
                    printf("%s%.4f", d ? ", " : "", best);
                }
                printf("]}\n");
                fflush(stdout);
            }
    }
    // the 80 MiB plan over a whole 128 MiB period of D - S
    {
        printf("{\"memory\": \"%s\", \"plan\": \"%s\", \"ms_by_D_minus_S_step_8MiB\": [", contiguous ? "contiguous" : "hipMalloc",
               plans[1].name);
        for (int d = 0; d < 20; d++) {
            float best = 1e30f;
            for (int rnd = 0; rnd < 2; rnd++)
                best = std::min(best, time_copy(arena, arena + first + (size_t)d * 8 * MiB, bytes, plans[1].plan, 8));
            printf("%s%.4f", d ? ", " : "", best);
        }
        printf("]}\n");
    }
    return 0;
}

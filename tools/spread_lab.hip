// spread_lab: VRAM seems to come in two CLASSES of region (region_lab: a copy between buffers of different classes
// runs at ~0.43 ms per 1.296 GB, inside one class at ~0.455; hipMalloc buffers that mix both are written faster).
// This lab classifies 1 GiB groups of separately created physical chunks by timing, then builds rasters whose
// chunks ALTERNATE between the classes and times every combination.  (GPU only; lab.)
//   hipcc -O3 --offload-arch=gfx950 -o tools/spread_lab tools/spread_lab.hip && tools/spread_lab [chunk MiB=32] [groups=32]
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
constexpr size_t GiB = 1ull << 30, MiB = 1ull << 20;

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
static float time_it(const void *src, void *dst, size_t bytes = GiB)
{
    uint32_t nvec = (uint32_t)(bytes / 16);
    uint32_t ntrips = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    const u32x4 *in = (const u32x4 *)src;
    u32x4 *out = (u32x4 *)dst;
    void *args[] = { &in, &out, &nvec, &ntrips };
    float best = 1e30f;
    for (int rnd = 0; rnd < 2; rnd++) {
        hipLaunchKernelGGL(sweep<MODE>, dim3(2048), dim3(kThreads), 0, 0, in, out, nvec, ntrips);
        for (int k = 0; k < 5; k++)
            CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(sweep<MODE>), dim3(2048), dim3(kThreads), args, 0, 0,
                                     e0[k], e1[k], 0));
        CHECK(hipDeviceSynchronize());
        float ms[5];
        for (int k = 0; k < 5; k++)
            CHECK(hipEventElapsedTime(&ms[k], e0[k], e1[k]));
        std::sort(ms, ms + 5);
        best = std::min(best, ms[2]);
    }
    return best;
}

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;

static char *reserve(size_t bytes)
{
    char *va = nullptr;
    CHECK(hipMemAddressReserve((void **)&va, bytes, 0, nullptr, 0));
    return va;
}

int main(int argc, char **argv)
{
    size_t chunk = (argc > 1 ? atoi(argv[1]) : 32) * MiB;
    int ngroups = argc > 2 ? atoi(argv[2]) : 32;
    const int per = (int)(GiB / chunk);
    for (int k = 0; k < 5; k++) {
        CHECK(hipEventCreate(&e0[k]));
        CHECK(hipEventCreate(&e1[k]));
    }
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    char *ref = nullptr;
    CHECK(hipExtMallocWithFlags((void **)&ref, GiB, hipDeviceMallocContiguous));
    CHECK(hipMemset(ref, 1, GiB));

    // groups of separately created chunks, each group mapped in creation order
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> h(ngroups, std::vector<hipMemGenericAllocationHandle_t>(per));
    std::vector<char *> gva(ngroups);
    std::vector<float> t_from_ref(ngroups), t_write(ngroups), t_read(ngroups);
    for (int g = 0; g < ngroups; g++) {
        gva[g] = reserve(GiB);
        for (int j = 0; j < per; j++) {
            CHECK(hipMemCreate(&h[g][j], chunk, &prop, 0));
            CHECK(hipMemMap(gva[g] + (size_t)j * chunk, chunk, 0, h[g][j], 0));
        }
        CHECK(hipMemSetAccess(gva[g], GiB, &acc, 1));
        CHECK(hipMemset(gva[g], g + 2, GiB));
    }
    for (int g = 0; g < ngroups; g++) {
        t_from_ref[g] = time_it<0>(ref, gva[g]);
        t_write[g] = time_it<2>(ref, gva[g]);
        t_read[g] = time_it<1>(gva[g], gva[g]);
    }
    auto print = [&](const char *what, const std::vector<float> &v) {
        printf("{\"chunk_MiB\": %zu, \"what\": \"%s\", \"ms\": [", chunk / MiB, what);
        for (size_t i = 0; i < v.size(); i++)
            printf("%s%.4f", i ? ", " : "", v[i]);
        printf("]}\n");
        fflush(stdout);
    };
    print("copy of 1 GiB: contiguous reference -> group g", t_from_ref);
    print("write-only sweep of group g", t_write);
    print("read-only sweep of group g", t_read);

    // two classes by the copy time from the reference
    float lo = *std::min_element(t_from_ref.begin(), t_from_ref.end());
    float hi = *std::max_element(t_from_ref.begin(), t_from_ref.end());
    float mid = 0.5f * (lo + hi);
    std::vector<int> ca, cb;        // ca: slow from the reference (same class as it), cb: fast (the other class)
    for (int g = 0; g < ngroups; g++) {
        // keep only clear members
        if (t_from_ref[g] > mid + 0.25f * (hi - mid))
            ca.push_back(g);
        else if (t_from_ref[g] < mid - 0.25f * (mid - lo))
            cb.push_back(g);
    }
    printf("{\"what\": \"classes\", \"spread_ms\": [%.4f, %.4f], \"same_class_as_reference\": %zu, \"other_class\": %zu}\n", lo, hi,
           ca.size(), cb.size());
    if (hi - lo < 0.01f || ca.size() < 2 || cb.size() < 2) {
        printf("{\"what\": \"no two clear classes among these groups; nothing more to measure\"}\n");
        return 0;
    }
    const int a1 = ca[0], a2 = ca[1], b1 = cb[0], b2 = cb[1];
    // the pairwise picture between whole groups
    {
        std::vector<float> v = { time_it<0>(gva[a1], gva[a2]), time_it<0>(gva[b1], gva[b2]), time_it<0>(gva[a1], gva[b2]),
                                 time_it<0>(gva[b1], gva[a2]) };
        print("whole groups: A->A, B->B, A->B, B->A", v);
    }
    // composite rasters: chunk j comes from group X if ((j / run) & 1) == phase else from group Y
    auto compose = [&](int gx, int gy, int run, int phase, int half) -> char * {
        // uses chunks [half*per/2, half*per/2 + per/2) of each group, so two composites can be built from one pair
        char *va = reserve(GiB);
        int ix = half * per / 2, iy = half * per / 2;
        for (int j = 0; j < per; j++) {
            bool x = ((j / run) & 1) == phase;
            hipMemGenericAllocationHandle_t hh = x ? h[gx][ix++] : h[gy][iy++];
            hipError_t e = hipMemMap(va + (size_t)j * chunk, chunk, 0, hh, 0);
            if (e != hipSuccess) {
                fprintf(stderr, "mapping a chunk a second time refused: %s\n", hipGetErrorString(e));
                exit(4);
            }
        }
        CHECK(hipMemSetAccess(va, GiB, &acc, 1));
        return va;
    };
    for (int run = 1; run <= per / 2; run *= 2) {
        // sources from (a1, b1), destinations from (a2, b2)
        char *s0 = compose(a1, b1, run, 0, 0);          // A first
        char *d0 = compose(a2, b2, run, 0, 0);          // A first: reads and writes of one offset in the SAME class
        char *d1 = compose(a2, b2, run, 1, 1);          // B first: reads and writes of one offset in DIFFERENT classes
        std::vector<float> v = { time_it<1>(s0, s0), time_it<2>(s0, d0), time_it<0>(s0, d0), time_it<0>(s0, d1),
                                 time_it<0>(gva[a1], d0), time_it<0>(gva[b1], d0), time_it<0>(s0, gva[a2]),
                                 time_it<0>(s0, gva[b2]) };
        char name[256];
        snprintf(name, sizeof name,
                 "alternating runs of %d chunk(s) (%zu MiB): read mix, write mix, mix->mix same phase, mix->mix opposite "
                 "phase, A->mix, B->mix, mix->A, mix->B", run, run * chunk / MiB);
        print(name, v);
    }
    return 0;
}

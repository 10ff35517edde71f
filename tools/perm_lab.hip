// perm_lab: in one PHYSICALLY CONTIGUOUS arena (relative physical addresses known), copy 1 GiB in 2 MiB chunks
// where chunk c of the source region goes to chunk map(c) of the destination region.  map = identity is a plain
// copy; other maps emulate scattered physical layouts with a KNOWN structure, to see which relation between the
// physical addresses read and written at the same time decides the copy rate.  (GPU only; lab.)
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/perm_lab tools/perm_lab.hip && tools/perm_lab
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
constexpr uint32_t kChunkTrips = 256;                   // 256 trips x 8 KiB = 2 MiB
constexpr uint32_t kChunks = 512;                       // 1 GiB

__global__ __launch_bounds__(kThreads) void perm_copy(const u32x4 *in, u32x4 *out, const uint16_t *smap,
                                                      const uint16_t *dmap)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7u;
    const uint32_t ntrips = kChunks * kChunkTrips;
    const uint32_t per = ntrips / 8u;
    const uint32_t lo = x * per, hi = lo + per;
    for (uint32_t trip = lo + (b >> 3); trip < hi; trip += nb / 8u) {
        const uint32_t c = trip / kChunkTrips, w = trip % kChunkTrips;
        const uint32_t st = (uint32_t)smap[c] * kChunkTrips + w;
        const uint32_t dt = (uint32_t)dmap[c] * kChunkTrips + w;
        u32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
            v[u] = __builtin_nontemporal_load(in + (st * 2u + u) * (uint32_t)kThreads + threadIdx.x);
#pragma unroll
        for (int u = 0; u < 2; u++)
            __builtin_nontemporal_store(v[u], out + (dt * 2u + u) * (uint32_t)kThreads + threadIdx.x);
    }
}

static hipEvent_t e0[5], e1[5];
static uint16_t *d_smap, *d_dmap;

static float run(const char *in, char *out, const std::vector<uint16_t> &smap, const std::vector<uint16_t> &dmap)
{
    CHECK(hipMemcpy(d_smap, smap.data(), kChunks * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dmap, dmap.data(), kChunks * 2, hipMemcpyHostToDevice));
    const u32x4 *i4 = (const u32x4 *)in;
    u32x4 *o4 = (u32x4 *)out;
    void *args[] = { &i4, &o4, &d_smap, &d_dmap };
    float best = 1e30f;
    for (int rnd = 0; rnd < 2; rnd++) {
        hipLaunchKernelGGL(perm_copy, dim3(2048), dim3(kThreads), 0, 0, i4, o4, d_smap, d_dmap);
        for (int k = 0; k < 5; k++)
            CHECK(hipExtLaunchKernel(reinterpret_cast<const void *>(perm_copy), dim3(2048), dim3(kThreads), args, 0, 0,
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

static std::vector<uint16_t> identity()
{
    std::vector<uint16_t> m(kChunks);
    for (uint32_t i = 0; i < kChunks; i++)
        m[i] = (uint16_t)i;
    return m;
}

static uint64_t rng_state = 0x1234567ull;
static uint32_t rnd32()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

int main(int argc, char **argv)
{
    bool contiguous = !(argc > 1 && atoi(argv[1]) == 0);
    for (int k = 0; k < 5; k++) {
        CHECK(hipEventCreate(&e0[k]));
        CHECK(hipEventCreate(&e1[k]));
    }
    const size_t GiB = 1ull << 30;
    char *arena = nullptr;
    if (contiguous)
        CHECK(hipExtMallocWithFlags((void **)&arena, 4 * GiB, hipDeviceMallocContiguous));
    else
        CHECK(hipMalloc((void **)&arena, 4 * GiB));
    CHECK(hipMemset(arena, 0x5a, 4 * GiB));
    CHECK(hipMalloc((void **)&d_smap, kChunks * 2));
    CHECK(hipMalloc((void **)&d_dmap, kChunks * 2));
    const char *src = arena;
    const char *mem = contiguous ? "contiguous" : "hipMalloc";
    auto id = identity();

    // destination region at several distances
    for (int gd = 1; gd <= 3; gd++)
        printf("{\"memory\": \"%s\", \"map\": \"identity\", \"dst_at_GiB\": %d, \"ms\": %.4f}\n", mem, gd,
               run(src, arena + gd * GiB, id, id));
    fflush(stdout);
    char *dst = arena + 2 * GiB;
    // one address bit of the destination (or source) chunk index flipped
    for (int side = 0; side < 2; side++) {
        printf("{\"memory\": \"%s\", \"map\": \"chunk index XOR (1 << bit) on the %s side\", \"ms_by_bit_0_to_8\": [", mem,
               side ? "source" : "destination");
        for (int bit = 0; bit < 9; bit++) {
            auto m = id;
            for (auto &v : m)
                v ^= (uint16_t)(1u << bit);
            printf("%s%.4f", bit ? ", " : "", side ? run(src, dst, m, id) : run(src, dst, id, m));
        }
        printf("]}\n");
        fflush(stdout);
    }
    // XOR with every mask of the low 7 bits (destination side)
    {
        printf("{\"memory\": \"%s\", \"map\": \"destination chunk index XOR mask\", \"ms_by_mask_0_to_127\": [", mem);
        for (int mask = 0; mask < 128; mask++) {
            auto m = id;
            for (auto &v : m)
                v ^= (uint16_t)mask;
            printf("%s%.4f", mask ? ", " : "", run(src, dst, id, m));
        }
        printf("]}\n");
        fflush(stdout);
    }
    // rotations: destination chunk = (c + r) mod 512  (same as moving the destination by r chunks, wrapped)
    {
        printf("{\"memory\": \"%s\", \"map\": \"destination chunk index + r (mod 512)\", \"ms_by_r_0_to_71\": [", mem);
        for (int r = 0; r < 72; r++) {
            auto m = id;
            for (auto &v : m)
                v = (uint16_t)((v + r) % kChunks);
            printf("%s%.4f", r ? ", " : "", run(src, dst, id, m));
        }
        printf("]}\n");
        fflush(stdout);
    }
    // random permutations: whole, within 128 MiB groups (low 6 bits), of the groups (high bits)
    for (int kind = 0; kind < 4; kind++) {
        static const char *names[] = { "random permutation of all 512 chunks (destination)",
                                       "random within each 128 MiB group (destination)",
                                       "random order of the eight 128 MiB groups (destination)",
                                       "random permutation on both sides" };
        printf("{\"memory\": \"%s\", \"map\": \"%s\", \"ms_by_seed\": [", mem, names[kind]);
        for (int seed = 0; seed < 6; seed++) {
            auto shuffle = [&](std::vector<uint16_t> &m, uint32_t lo, uint32_t n) {
                for (uint32_t i = n - 1; i > 0; i--)
                    std::swap(m[lo + i], m[lo + rnd32() % (i + 1)]);
            };
            auto m = id, s = id;
            if (kind == 0 || kind == 3)
                shuffle(m, 0, kChunks);
            if (kind == 3)
                shuffle(s, 0, kChunks);
            if (kind == 1)
                for (uint32_t g = 0; g < kChunks; g += 64)
                    shuffle(m, g, 64);
            if (kind == 2) {
                std::vector<uint16_t> g(8);
                for (int i = 0; i < 8; i++)
                    g[i] = (uint16_t)i;
                for (int i = 7; i > 0; i--)
                    std::swap(g[i], g[rnd32() % (i + 1)]);
                for (uint32_t i = 0; i < kChunks; i++)
                    m[i] = (uint16_t)(g[i / 64] * 64 + i % 64);
            }
            printf("%s%.4f", seed ? ", " : "", run(src, dst, s, m));
        }
        printf("]}\n");
        fflush(stdout);
    }
    return 0;
}

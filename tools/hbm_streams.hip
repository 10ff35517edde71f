// hbm_streams.hip -- measures the streaming ceilings of this MI355X for the
// read:write mixes of the CN kernels (1:1, 1:9, 1:18) over launch shapes
// (workgroup->address mapping, loads in flight per lane, grid size).
// Diagnostic only; prints one JSON line per variant.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_streams tools/hbm_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Outs { u32x4 *p[18]; };

// MAP 0: grid-stride, consecutive workgroups touch consecutive 4 KiB chunks
// MAP 1: XCD slabs: workgroups with equal blockIdx%8 stream one contiguous eighth
// MAP 2: every workgroup streams its own contiguous range
template <int NW, int UN, int MAP>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4 *in, Outs o, size_t nvec)
{
    const size_t nchunk = (nvec + 256 * UN - 1) / (256 * UN);   // chunk = 256*UN vectors
    size_t c, cend, cstep;
    if (MAP == 0) { c = blockIdx.x; cend = nchunk; cstep = gridDim.x; }
    else if (MAP == 1) {
        size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
        c = xcd * per + (blockIdx.x >> 3); cend = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk; cstep = gridDim.x / 8;
    }
    else {
        size_t per = (nchunk + gridDim.x - 1) / gridDim.x;
        c = blockIdx.x * per; cend = c + per < nchunk ? c + per : nchunk; cstep = 1;
    }
    for (; c < cend; c += cstep) {
        u32x4 v[UN];
        size_t idx[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            idx[u] = (c * UN + u) * 256 + threadIdx.x;
            if (idx[u] < nvec) v[u] = __builtin_nontemporal_load(in + idx[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            if (idx[u] >= nvec) continue;
#pragma unroll
            for (int k = 0; k < NW; k++)
                __builtin_nontemporal_store(v[u] + (uint32_t)k, o.p[k] + idx[u]);
        }
    }
}

// variant: every lane owns UN *consecutive* 16-byte vectors (32 or 64 contiguous bytes per lane)
template <int NW, int UN>
__global__ __launch_bounds__(256) void stream_kernel_contig(const u32x4 *in, Outs o, size_t nvec)
{
    const size_t nchunk = (nvec + 256 * UN - 1) / (256 * UN);
    const size_t per = (nchunk + 7) / 8, xcd = blockIdx.x & 7;
    size_t c = xcd * per + (blockIdx.x >> 3);
    const size_t cend = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk, cstep = gridDim.x / 8;
    for (; c < cend; c += cstep) {
        u32x4 v[UN];
        const size_t base = (c * 256 + threadIdx.x) * UN;
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (base + u < nvec) v[u] = __builtin_nontemporal_load(in + base + u);
#pragma unroll
        for (int u = 0; u < UN; u++) {
            if (base + u >= nvec) continue;
#pragma unroll
            for (int k = 0; k < NW; k++)
                __builtin_nontemporal_store(v[u] + (uint32_t)k, o.p[k] + base + u);
        }
    }
}

template <int NW, int UN>
void run_contig(const u32x4 *in, Outs o, size_t bytes, int blocks)
{
    size_t nvec = bytes / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stream_kernel_contig<NW, UN>), dim3(blocks), dim3(256), 0, 0, in, o, nvec);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((stream_kernel_contig<NW, UN>), dim3(blocks), dim3(256), 0, 0, in, o, nvec);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double gb = (double)bytes * (1 + NW) / 1e9;
    printf("{\"mix\": \"1R:%dW\", \"unroll\": %d, \"map\": \"contig\", \"blocks\": %d, \"ms\": %.4f, \"GBps\": %.1f, \"frac_of_8TBps\": %.4f}\n",
           NW, UN, blocks, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000.0);
    fflush(stdout);
}

template <int NW, int UN, int MAP>
void run(const u32x4 *in, Outs o, size_t bytes, int blocks)
{
    size_t nvec = bytes / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stream_kernel<NW, UN, MAP>), dim3(blocks), dim3(256), 0, 0, in, o, nvec);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((stream_kernel<NW, UN, MAP>), dim3(blocks), dim3(256), 0, 0, in, o, nvec);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double gb = (double)bytes * (1 + NW) / 1e9;
    printf("{\"mix\": \"1R:%dW\", \"unroll\": %d, \"map\": %d, \"blocks\": %d, \"ms\": %.4f, \"GBps\": %.1f, \"frac_of_8TBps\": %.4f}\n",
           NW, UN, MAP, blocks, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000.0);
    fflush(stdout);
}

template <int NW>
void sweep(const u32x4 *in, Outs o, size_t bytes)
{
    int blks[] = {1024, 2048, 4096, 8192};
    for (int b : blks) {
        run<NW, 1, 0>(in, o, bytes, b); run<NW, 1, 1>(in, o, bytes, b); run<NW, 1, 2>(in, o, bytes, b);
        run<NW, 2, 0>(in, o, bytes, b); run<NW, 2, 1>(in, o, bytes, b); run<NW, 2, 2>(in, o, bytes, b);
        run<NW, 4, 0>(in, o, bytes, b); run<NW, 4, 1>(in, o, bytes, b);
    }
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)36000 * 36000;     // one CN raster
    u32x4 *in; Outs o;
    CK(hipMalloc((void **)&in, bytes));
    CK(hipMemset(in, 1, bytes));
    for (int k = 0; k < 18; k++) CK(hipMalloc((void **)&o.p[k], bytes));
    sweep<1>(in, o, bytes);
    for (int b : {1024, 2048, 4096, 8192}) {
        run_contig<1, 2>(in, o, bytes, b);
        run_contig<1, 4>(in, o, bytes, b);
    }
    sweep<18>(in, o, bytes);
    if (argc > 1) sweep<9>(in, o, bytes);
    return 0;
}

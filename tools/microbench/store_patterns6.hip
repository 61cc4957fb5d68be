// store_patterns6.hip — does the write rate of the render pattern depend on WHICH buffer it writes?
// The same binary rasters at 0.155 / 0.18 / 0.20 ms from one process to the next on one box while a fill is
// constant.  Hypothesis: the physical placement (fragment size / page contiguity) of the 932 MB obs buffer —
// a scattered pattern (4096 waves, each streaming its own region) keeps thousands of pages live, a dense
// front only a few.  Test: 8 buffers of that size allocated one after the other in one process (all kept),
// fill / dense front / render pattern on each, twice.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns6.hip -o store_patterns6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int REGION = 28224, RCH = REGION / 16;

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * 256;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += per_round) out[c] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(1024) void k_wave_run(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (nregions + gridDim.x * 16 - 1) / (gridDim.x * 16);
    const int e0 = (blockIdx.x * 16 + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    for (int e = e0; e < e1; e++) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + 192 < RCH; c += 256) { o[c] = make_uint4(e, c, 0, 4); o[c + 64] = make_uint4(e, c, 1, 4); o[c + 128] = make_uint4(e, c, 2, 4); o[c + 192] = make_uint4(e, c, 3, 4); }
        for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 8;
    const int nregions = 32768;
    const size_t bytes = (size_t)nregions * REGION, nch = bytes / 16;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 40; i++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return ms / 40;
    };
    std::vector<uint4*> bufs;
    // a little fragmentation on purpose between the big allocations, like a real process has
    std::vector<void*> small;
    for (int i = 0; i < NB; i++) {
        uint4* p; CK(hipMalloc(&p, bytes));
        bufs.push_back(p);
        for (int j = 0; j < 3; j++) { void* q; CK(hipMalloc(&q, (size_t)(1 + j) << 20)); small.push_back(q); }
    }
    for (int rep = 0; rep < 2; rep++)
        for (int i = 0; i < NB; i++) {
            uint4* p = bufs[i];
            const float f = timeit([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, p, nch); });
            const float d = timeit([&] { hipLaunchKernelGGL(k_front, dim3(256), dim3(256), 0, 0, p, nch); });
            const float r = timeit([&] { hipLaunchKernelGGL(k_wave_run, dim3(256), dim3(1024), 0, 0, p, nregions); });
            printf("rep %d buffer %d @ %p (mod 2MiB %zu): fill %5.0f  dense front %5.0f  render pattern %5.0f GB/s\n", rep, i, (void*)p,
                   (size_t)((uintptr_t)p % (2u << 20)), bytes / f / 1e6, bytes / d / 1e6, bytes / r / 1e6);
            fflush(stdout);
        }
    return 0;
}

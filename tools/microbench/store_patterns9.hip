// store_patterns9.hip — at which STREAM SPACING does a badly placed buffer start to hurt?  256 x W waves (W per CU),
// each streaming `spacing` contiguous bytes in 4 KiB rounds; the buffer is covered block by block (a block = all
// the waves' streams side by side), so spacing = 4 KiB is the chip-wide dense front and spacing = buffer / waves
// the per-wave streams of the render pattern — same number of waves and stores in flight throughout.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns9.hip -o store_patterns9
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int REGION = 28224, RCH = REGION / 16;

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
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
// spacing_ch: chunks (16 B) per stream, a multiple of 256 (4 KiB)
__global__ __launch_bounds__(1024) void k_spaced(uint4* out, size_t nchunks, int W, size_t spacing_ch) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * W + (threadIdx.x >> 6), waves = (size_t)gridDim.x * W;
    const size_t block_ch = waves * spacing_ch;
    for (size_t b0 = 0; b0 < nchunks; b0 += block_ch) {
        const size_t s0 = b0 + w * spacing_ch, s1 = min(nchunks, s0 + spacing_ch);
        for (size_t c = s0 + lane; c < s1; c += 256) {
            out[c] = make_uint4(1, 0, 0, 4);
            if (c + 64 < s1) out[c + 64] = make_uint4(1, 1, 0, 4);
            if (c + 128 < s1) out[c + 128] = make_uint4(1, 2, 0, 4);
            if (c + 192 < s1) out[c + 192] = make_uint4(1, 3, 0, 4);
        }
    }
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 24;
    const int nregions = 32768;
    const size_t bytes = (size_t)nregions * REGION, nch = bytes / 16;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](auto launch, int n = 30) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < n; i++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return ms / n;
    };
    std::vector<uint4*> bufs;
    std::vector<std::pair<float, int>> rate;
    for (int i = 0; i < NB; i++) {
        uint4* p; CK(hipMalloc(&p, bytes));
        bufs.push_back(p);
        const float r = timeit([&] { hipLaunchKernelGGL(k_wave_run, dim3(256), dim3(1024), 0, 0, p, nregions); }, 10);
        rate.push_back({(float)(bytes / r / 1e6), i});
    }
    std::sort(rate.begin(), rate.end());
    printf("render pattern over %d buffers: min %.0f median %.0f max %.0f GB/s\n", NB, rate.front().first, rate[NB / 2].first, rate.back().first);
    const int pick[2] = {rate.front().second, rate.back().second};
    const char* cls[2] = {"SLOWEST", "FASTEST"};
    for (int k = 0; k < 2; k++) {
        uint4* p = bufs[pick[k]];
        printf("--- %s buffer (%d) @ %p\n", cls[k], pick[k], (void*)p);
        auto show = [&](const char* name, float ms) { printf("%-64s %.4f ms %5.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
        show("fill", timeit([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, p, nch); }));
        for (int W : {1, 4, 16})
        for (size_t kib : {4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384}) {
            if ((size_t)256 * W * kib * 1024 > bytes * 2) continue;
            char name[128];
            snprintf(name, sizeof name, "%2d waves/CU, stream spacing %6zu KiB (window %7.1f MiB)", W, kib, 256.0 * W * kib / 1024);
            show(name, timeit([&] { hipLaunchKernelGGL(k_spaced, dim3(256), dim3(W * 64), 0, 0, p, nch, W, kib * 64); }));
        }
    }
    return 0;
}

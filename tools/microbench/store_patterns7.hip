// store_patterns7.hip — how many waves have to form a DENSE, ALIGNED write front before a badly placed buffer
// stops mattering?  store_patterns6 showed: the render pattern (4096 waves, each streaming its own run of
// regions) is absorbed at 5.3-5.7 TB/s by most obs-sized allocations and at 6.8 TB/s by a few, while a chip-wide
// dense front (F256) runs at 6.6-6.8 TB/s on all of them.  Here the family in between: the 4096 waves of a
// render-shaped launch (256 workgroups x 16 waves) in groups of G consecutive waves; a group owns one contiguous,
// 4 KiB-aligned span of the buffer and sweeps it as a dense front — per round every wave stores 4 KiB (4 x 1 KiB
// store instructions).  G = 1 is the render pattern with aligned stores, G = 16 a workgroup-wide front,
// G = 4096 the chip-wide front.  MODE 0: a wave's 4 KiB are contiguous; MODE 1: its four 1 KiB stores are
// interleaved with the other waves' (the front advances 1 KiB x G per store instruction).
// Run on the slowest, the median and the fastest of NB freshly allocated buffers (classified by the render pattern).
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns7.hip -o store_patterns7
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
// groups of G consecutive waves (wave id = blockIdx * 16 + wave-in-block), one dense front per group
template <int MODE>
__global__ __launch_bounds__(1024) void k_group_front(uint4* out, size_t nchunks, int G, size_t span) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
    const int grp = w / G, j = w - grp * G;
    const size_t s0 = (size_t)grp * span, s1 = min(nchunks, s0 + span);
    if (MODE == 0) {
        for (size_t c = s0 + (size_t)j * 256 + lane; c < s1; c += (size_t)G * 256) {
            out[c] = make_uint4(w, 0, 0, 4);
            if (c + 64 < s1) out[c + 64] = make_uint4(w, 1, 0, 4);
            if (c + 128 < s1) out[c + 128] = make_uint4(w, 2, 0, 4);
            if (c + 192 < s1) out[c + 192] = make_uint4(w, 3, 0, 4);
        }
    } else {
        const size_t GS = (size_t)G * 64;
        for (size_t c = s0 + (size_t)j * 64 + lane; c < s1; c += 4 * GS) {
            out[c] = make_uint4(w, 0, 0, 4);
            if (c + GS < s1) out[c + GS] = make_uint4(w, 1, 0, 4);
            if (c + 2 * GS < s1) out[c + 2 * GS] = make_uint4(w, 2, 0, 4);
            if (c + 3 * GS < s1) out[c + 3 * GS] = make_uint4(w, 3, 0, 4);
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
        printf("buffer %2d @ %p render pattern %5.0f GB/s\n", i, (void*)p, bytes / r / 1e6);
    }
    std::sort(rate.begin(), rate.end());
    const int pick[3] = {rate.front().second, rate[NB / 2].second, rate.back().second};
    const char* cls[3] = {"SLOWEST", "MEDIAN", "FASTEST"};
    for (int rep = 0; rep < 2; rep++)
    for (int k = 0; k < 3; k++) {
        uint4* p = bufs[pick[k]];
        printf("--- %s buffer (%d) rep %d\n", cls[k], pick[k], rep);
        auto show = [&](const char* name, float ms) { printf("%-58s %.4f ms %5.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
        show("fill", timeit([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, p, nch); }));
        show("F256 chip-wide front, 1024 waves x 1 KiB", timeit([&] { hipLaunchKernelGGL(k_front, dim3(256), dim3(256), 0, 0, p, nch); }));
        show("render pattern (wave per region, misaligned)", timeit([&] { hipLaunchKernelGGL(k_wave_run, dim3(256), dim3(1024), 0, 0, p, nregions); }));
        const int Gs[] = {1, 2, 4, 16, 64, 256, 1024, 4096};
        for (int G : Gs) {
            const int ngroups = 4096 / G;
            const size_t span = ((nch + ngroups - 1) / ngroups + 255) / 256 * 256;
            char name[128];
            snprintf(name, sizeof name, "group front G = %4d, wave writes 4 KiB contiguous", G);
            show(name, timeit([&] { hipLaunchKernelGGL((k_group_front<0>), dim3(256), dim3(1024), 0, 0, p, nch, G, span); }));
            if (G > 1) {
                snprintf(name, sizeof name, "group front G = %4d, 1 KiB stores interleaved", G);
                show(name, timeit([&] { hipLaunchKernelGGL((k_group_front<1>), dim3(256), dim3(1024), 0, 0, p, nch, G, span); }));
            }
        }
    }
    return 0;
}

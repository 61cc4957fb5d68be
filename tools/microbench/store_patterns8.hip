// store_patterns8.hip — follow-up to store_patterns7: on a badly placed buffer even a chip-wide dense front is slow
// when 4096 waves x 4 KiB are in flight (5.5 TB/s), while F256 — 1024 waves, 1 KiB each — runs at 6.6 TB/s.  So:
// is it the NUMBER OF WAVES / BYTES IN FLIGHT?  Per-wave streams (G = 1, the render pattern with aligned stores)
// and chip-wide fronts (G = all), with W = 1, 2, 4, 8, 16 waves per CU and S = 1, 2, 4 stores of 1 KiB per round.
// On the slowest and the fastest of NB freshly allocated buffers.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns8.hip -o store_patterns8
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
// W waves per workgroup, one workgroup per CU; groups of G consecutive waves sweep a span as a dense front,
// S stores of 1 KiB per wave and round (a wave's S KiB contiguous)
template <int S>
__global__ __launch_bounds__(1024) void k_front(uint4* out, size_t nchunks, int W, int G, size_t span) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * W + (threadIdx.x >> 6);
    const int grp = w / G, j = w - grp * G;
    const size_t s0 = (size_t)grp * span, s1 = min(nchunks, s0 + span);
    for (size_t c = s0 + (size_t)j * (64 * S) + lane; c < s1; c += (size_t)G * (64 * S)) {
#pragma unroll
        for (int q = 0; q < S; q++) if (c + q * 64 < s1) out[c + q * 64] = make_uint4(w, q, 0, 4);
    }
}
// the same with a pause of `idle` s_sleep units after every round (a wave that alternates storing with other work)
template <int S>
__global__ __launch_bounds__(1024) void k_front_idle(uint4* out, size_t nchunks, int W, int G, size_t span, int idle) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * W + (threadIdx.x >> 6);
    const int grp = w / G, j = w - grp * G;
    const size_t s0 = (size_t)grp * span, s1 = min(nchunks, s0 + span);
    for (size_t c = s0 + (size_t)j * (64 * S) + lane; c < s1; c += (size_t)G * (64 * S)) {
#pragma unroll
        for (int q = 0; q < S; q++) if (c + q * 64 < s1) out[c + q * 64] = make_uint4(w, q, 0, 4);
        for (int i = 0; i < idle; i++) __builtin_amdgcn_s_sleep(8);
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
        printf("--- %s buffer (%d)\n", cls[k], pick[k]);
        auto show = [&](const char* name, float ms) { printf("%-64s %.4f ms %5.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
        show("fill", timeit([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, p, nch); }));
        show("render pattern (16 waves/CU, wave per region, misaligned)", timeit([&] { hipLaunchKernelGGL(k_wave_run, dim3(256), dim3(1024), 0, 0, p, nregions); }));
        for (int all = 0; all < 2; all++)
        for (int W : {1, 2, 4, 8, 16})
        for (int S : {1, 2, 4, 8}) {
            const int waves = 256 * W, G = all ? waves : 1, ngroups = waves / G;
            const size_t span = ((nch + ngroups - 1) / ngroups + 64 * S - 1) / (64 * S) * (64 * S);
            char name[128];
            snprintf(name, sizeof name, "%s, %2d waves/CU, %d KiB per wave and round", all ? "chip-wide front  " : "per-wave streams ", W, S);
            float ms = 0;
            if (S == 1) ms = timeit([&] { hipLaunchKernelGGL((k_front<1>), dim3(256), dim3(W * 64), 0, 0, p, nch, W, G, span); });
            if (S == 2) ms = timeit([&] { hipLaunchKernelGGL((k_front<2>), dim3(256), dim3(W * 64), 0, 0, p, nch, W, G, span); });
            if (S == 4) ms = timeit([&] { hipLaunchKernelGGL((k_front<4>), dim3(256), dim3(W * 64), 0, 0, p, nch, W, G, span); });
            if (S == 8) ms = timeit([&] { hipLaunchKernelGGL((k_front<8>), dim3(256), dim3(W * 64), 0, 0, p, nch, W, G, span); });
            show(name, ms);
        }
        // 16 waves per CU that pause between rounds: fewer of them are storing at any time
        for (int idle : {1, 2, 4, 8, 16}) {
            const int W = 16, waves = 256 * W;
            const size_t span = ((nch + waves - 1) / waves + 255) / 256 * 256;
            char name[128];
            snprintf(name, sizeof name, "per-wave streams, 16 waves/CU, 4 KiB, then s_sleep(8) x %d", idle);
            show(name, timeit([&] { hipLaunchKernelGGL((k_front_idle<4>), dim3(256), dim3(W * 64), 0, 0, p, nch, W, 1, span, idle); }));
        }
    }
    return 0;
}

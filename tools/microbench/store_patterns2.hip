// store_patterns2.hip — pure-store HBM write patterns, measured HOT: after a sustained warm-up every
// pattern runs for ~0.4 s back to back, in several interleaved rounds, so clock / thermal drift shows
// up as drift across rounds instead of hiding in one-shot numbers.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns2.hip -o store_patterns2 ; run on the GPU box.
// All patterns write the bench workload's obs tensor: 32768 regions x 28224 B (3 x 56 x 56 x 3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 28224;
constexpr int RCH = REGION / 16;   // 1764 chunks of 16 B

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
// wave per region, grid-strided over regions, UNR stores back to back; WPB waves per workgroup
template <int WPB, int UNR>
__global__ __launch_bounds__(WPB * 64) void k_wave_strided(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * WPB + wave; e < nregions; e += gridDim.x * WPB) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + (UNR - 1) * 64 < RCH; c += UNR * 64)
#pragma unroll
            for (int u = 0; u < UNR; u++) o[c + u * 64] = make_uint4(e, c, u, 4);
        for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// wave per region, every wave walks its own CONTIGUOUS run of regions (the render kernel's pattern)
template <int WPB, int UNR>
__global__ __launch_bounds__(WPB * 64) void k_wave_run(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (nregions + gridDim.x * WPB - 1) / (gridDim.x * WPB);
    const int e0 = (blockIdx.x * WPB + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    for (int e = e0; e < e1; e++) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + (UNR - 1) * 64 < RCH; c += UNR * 64)
#pragma unroll
            for (int u = 0; u < UNR; u++) o[c + u * 64] = make_uint4(e, c, u, 4);
        for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// TEAM of TW waves per region (TW*1 KiB contiguous per store step), workgroup of WPB waves,
// grid-strided over regions
template <int WPB, int TW>
__global__ __launch_bounds__(WPB * 64) void k_team_strided(uint4* out, int nregions) {
    constexpr int TEAMS = WPB / TW, TT = TW * 64;
    const int team = threadIdx.x / TT, t = threadIdx.x % TT;
    for (int e = blockIdx.x * TEAMS + team; e < nregions; e += gridDim.x * TEAMS) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = t; c < RCH; c += TT) o[c] = make_uint4(e, c, 3, 4);
    }
}
// workgroup walks its own contiguous run of regions cooperatively, as one long stream of WPB KiB steps
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_block_run(uint4* out, int nregions) {
    const int per_wg = (nregions + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per_wg, e1 = min(nregions, e0 + per_wg);
    if (e0 >= e1) return;
    uint4* o = out + (size_t)e0 * RCH;
    const size_t tot = (size_t)(e1 - e0) * RCH;
    for (size_t c = threadIdx.x; c < tot; c += WPB * 64) o[c] = make_uint4(e0, (unsigned)c, 3, 4);
}
// dense moving front over the whole tensor: in round r workgroup w writes block (r * gridDim + w)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS;
    for (size_t c = (size_t)blockIdx.x * THREADS + threadIdx.x; c < nchunks; c += per_round) out[c] = make_uint4(1, 2, 3, 4);
}
// dense front over WINDOWS: the tensor is cut into gridDim/G windows... each group of G workgroups owns a
// contiguous window and sweeps it as a dense front of G*THREADS*16 B (locality between "whole chip" and "per wave")
template <int THREADS, int G>
__global__ __launch_bounds__(THREADS) void k_group_front(uint4* out, size_t nchunks) {
    const int ngroups = gridDim.x / G, grp = blockIdx.x / G, w = blockIdx.x % G;
    const size_t per_grp = (nchunks + ngroups - 1) / ngroups;
    const size_t c0 = (size_t)grp * per_grp, c1 = min(nchunks, c0 + per_grp);
    for (size_t c = c0 + (size_t)w * THREADS + threadIdx.x; c < c1; c += (size_t)G * THREADS) out[c] = make_uint4(1, 2, 3, 4);
}
// the raster's LDS side without its arithmetic: every store is fed by one ds_read_b128 from a linear
// LDS buffer (what a "rows assembled in LDS, then streamed" raster would do), wave per region run
template <int WPB, int UNR>
__global__ __launch_bounds__(WPB * 64) void k_wave_run_lds(uint4* out, int nregions) {
    __shared__ uint4 buf[WPB][UNR * 64 + 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < UNR * 64; i += 64) buf[wave][i] = make_uint4(i, wave, 3, 4);
    __syncthreads();
    const int per_wave = (nregions + gridDim.x * WPB - 1) / (gridDim.x * WPB);
    const int e0 = (blockIdx.x * WPB + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    for (int e = e0; e < e1; e++) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + (UNR - 1) * 64 < RCH; c += UNR * 64) {
            uint4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) v[u] = buf[wave][(lane + u * 64 + (c & 7)) % (UNR * 64)];
#pragma unroll
            for (int u = 0; u < UNR; u++) o[c + u * 64] = v[u];
        }
        for (; c < RCH; c += 64) o[c] = buf[wave][lane];
    }
}

struct Pattern { std::string name; std::function<void()> launch; };

int main(int argc, char** argv) {
    const float warm_s = argc > 1 ? atof(argv[1]) : 15.f;
    const float slice_s = argc > 2 ? atof(argv[2]) : 0.4f;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int nregions = 32768;
    const size_t bytes = (size_t)nregions * REGION, nch = bytes / 16;
    uint4* out; CK(hipMalloc(&out, bytes));
    std::vector<Pattern> P;
    P.push_back({"A   fill, 1 store/thread", [&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, out, nch); }});
    P.push_back({"L1  16-wave wg, wave/region strided x1, 256 wgs", [&] { hipLaunchKernelGGL((k_wave_strided<16, 1>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"L4  16-wave wg, wave/region strided x4, 256 wgs", [&] { hipLaunchKernelGGL((k_wave_strided<16, 4>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"R4  16-wave wg, wave/region RUN x4, 256 wgs (render pattern)", [&] { hipLaunchKernelGGL((k_wave_run<16, 4>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"R4b 4-wave wg, wave/region RUN x4, 1280 wgs", [&] { hipLaunchKernelGGL((k_wave_run<4, 4>), dim3(1280), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"R4c 4-wave wg, wave/region RUN x4, 512 wgs", [&] { hipLaunchKernelGGL((k_wave_run<4, 4>), dim3(512), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"R8  16-wave wg, wave/region RUN x8, 256 wgs", [&] { hipLaunchKernelGGL((k_wave_run<16, 8>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"E4  4-wave wg, wave/region strided x4, 1280 wgs", [&] { hipLaunchKernelGGL((k_wave_strided<4, 4>), dim3(1280), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"T2  4-wave wg, 2-wave team/region, 1280 wgs", [&] { hipLaunchKernelGGL((k_team_strided<4, 2>), dim3(1280), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"T4  4-wave wg, 4-wave team/region, 1280 wgs", [&] { hipLaunchKernelGGL((k_team_strided<4, 4>), dim3(1280), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"T4s 16-wave wg, 4-wave team/region, 256 wgs", [&] { hipLaunchKernelGGL((k_team_strided<16, 4>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"T16 16-wave wg, 16-wave team/region, 256 wgs", [&] { hipLaunchKernelGGL((k_team_strided<16, 16>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"B4  4-wave wg walks its own run cooperatively, 1024 wgs", [&] { hipLaunchKernelGGL((k_block_run<4>), dim3(1024), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"B16 16-wave wg walks its own run cooperatively, 256 wgs", [&] { hipLaunchKernelGGL((k_block_run<16>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"F256  dense front 256 thr x 256 wgs", [&] { hipLaunchKernelGGL((k_front<256>), dim3(256), dim3(256), 0, 0, out, nch); }});
    P.push_back({"F1024 dense front 1024 thr x 256 wgs", [&] { hipLaunchKernelGGL((k_front<1024>), dim3(256), dim3(1024), 0, 0, out, nch); }});
    P.push_back({"G8    8-wg group fronts, 256 thr x 1024 wgs", [&] { hipLaunchKernelGGL((k_group_front<256, 8>), dim3(1024), dim3(256), 0, 0, out, nch); }});
    P.push_back({"G32   32-wg group fronts, 256 thr x 1024 wgs", [&] { hipLaunchKernelGGL((k_group_front<256, 32>), dim3(1024), dim3(256), 0, 0, out, nch); }});
    P.push_back({"S4  R4 fed by ds_read_b128 (linear LDS -> HBM)", [&] { hipLaunchKernelGGL((k_wave_run_lds<16, 4>), dim3(256), dim3(1024), 0, 0, out, nregions); }});

    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run_for = [&](const Pattern& p, float seconds, float& ms_avg) {
        // launches in batches of 50 between events until `seconds` of GPU time has passed
        double total = 0; long n = 0;
        p.launch(); CK(hipDeviceSynchronize());
        while (total < seconds * 1e3) {
            CK(hipEventRecord(a));
            for (int i = 0; i < 50; i++) p.launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            total += ms; n += 50;
        }
        ms_avg = (float)(total / n);
    };
    float ms;
    printf("# cold first: one slice each, fresh GPU\n");
    for (auto& p : P) { run_for(p, 0.15f, ms); printf("cold  %-62s %.4f ms %6.0f GB/s\n", p.name.c_str(), ms, bytes / ms / 1e6); fflush(stdout); }
    printf("# warm-up: %.0f s of R4\n", warm_s);
    for (float t = 0; t < warm_s; t += 1.f) { run_for(P[3], 1.f, ms); printf("warm  t=%4.0fs R4 %.4f ms %6.0f GB/s\n", t + 1, ms, bytes / ms / 1e6); fflush(stdout); }
    for (int r = 0; r < rounds; r++)
        for (auto& p : P) { run_for(p, slice_s, ms); printf("hot%d  %-62s %.4f ms %6.0f GB/s\n", r, p.name.c_str(), ms, bytes / ms / 1e6); fflush(stdout); }
    CK(hipFree(out));
    return 0;
}

// store_patterns5.hip — ALIGNMENT of a wave's store instructions.  F256 (dense front) loses 22 % when its base
// is shifted by 64 B; the render kernel's regions are 28224 B = 441 x 64 B, so its 1 KiB wave stores sit at
// arbitrary 64-B phases.  Here: the render pattern (wave per region, contiguous run of regions per wave, 4 stores
// back to back) with the wave's chunks (a) counted from the region start, as the kernel does, (b) cut at
// 1 KiB-aligned global addresses (masked partial first / last store per region), (c) as (b) with every 4-store
// burst starting on a 4 KiB boundary, (d) the run as one stream cut at 1 KiB / 4 KiB boundaries (no per-region
// partial stores: what a stream-ordered raster could do).
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns5.hip -o store_patterns5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 28224;
constexpr int RCH = REGION / 16;

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS;
    for (size_t c = (size_t)blockIdx.x * THREADS + threadIdx.x; c < nchunks; c += per_round) out[c] = make_uint4(1, 2, 3, 4);
}
// MODE 0: chunks counted from the region start; 1: cut at 1 KiB-aligned addresses per region; 2: per region,
// bursts of 4 aligned to 4 KiB (head: single 1 KiB stores up to the boundary)
template <int WPB, int MODE>
__global__ __launch_bounds__(WPB * 64) void k_wave_run(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (nregions + gridDim.x * WPB - 1) / (gridDim.x * WPB);
    const int e0 = (blockIdx.x * WPB + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    for (int e = e0; e < e1; e++) {
        const long long g0 = (long long)e * RCH;                 // first chunk of the region (global chunk index)
        if (MODE == 0) {
            uint4* o = out + g0;
            int c = lane;
            for (; c + 192 < RCH; c += 256) { o[c] = make_uint4(e, c, 0, 4); o[c + 64] = make_uint4(e, c, 1, 4); o[c + 128] = make_uint4(e, c, 2, 4); o[c + 192] = make_uint4(e, c, 3, 4); }
            for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
        } else {
            const long long gend = g0 + RCH;
            long long g = (g0 & ~63LL) + lane;                    // 1 KiB-aligned: lane l always writes chunk = l mod 64
            if (MODE == 2) {                                      // singles until the next 4 KiB boundary
                for (; (g & 255LL & ~63LL) != 0 && (g & ~63LL) < gend; g += 64) if (g >= g0 && g < gend) out[g] = make_uint4(e, (int)g, 9, 4);
            }
            for (; (g & ~63LL) + 256 <= gend; g += 256) {
                if (g >= g0) out[g] = make_uint4(e, (int)g, 0, 4);   // only the very first store can start before g0
                out[g + 64] = make_uint4(e, (int)g, 1, 4); out[g + 128] = make_uint4(e, (int)g, 2, 4); out[g + 192] = make_uint4(e, (int)g, 3, 4);
            }
            for (; (g & ~63LL) < gend; g += 64) if (g >= g0 && g < gend) out[g] = make_uint4(e, (int)g, 8, 4);
        }
    }
}
// the wave's whole run as ONE stream cut at aligned addresses: ALIGN_CH = 64 (1 KiB) or 256 (4 KiB bursts)
template <int WPB, int BURST_ALIGN>
__global__ __launch_bounds__(WPB * 64) void k_wave_stream(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (nregions + gridDim.x * WPB - 1) / (gridDim.x * WPB);
    const int e0 = (blockIdx.x * WPB + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    if (e0 >= e1) return;
    const long long g0 = (long long)e0 * RCH, gend = (long long)e1 * RCH;
    long long g = (g0 & ~63LL) + lane;
    if (BURST_ALIGN) for (; (g & 255LL & ~63LL) != 0 && (g & ~63LL) < gend; g += 64) if (g >= g0 && g < gend) out[g] = make_uint4(1, (int)g, 9, 4);
    for (; (g & ~63LL) + 256 <= gend; g += 256) {
        if (g >= g0) out[g] = make_uint4(1, (int)g, 0, 4);
        out[g + 64] = make_uint4(1, (int)g, 1, 4); out[g + 128] = make_uint4(1, (int)g, 2, 4); out[g + 192] = make_uint4(1, (int)g, 3, 4);
    }
    for (; (g & ~63LL) < gend; g += 64) if (g >= g0 && g < gend) out[g] = make_uint4(1, (int)g, 8, 4);
}

struct Pattern { std::string name; std::function<void()> launch; };

int main(int argc, char** argv) {
    const float slice_s = argc > 1 ? atof(argv[1]) : 0.3f;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const int nregions = 32768;
    const size_t bytes = (size_t)nregions * REGION, nch = bytes / 16;
    uint4* out; CK(hipMalloc(&out, bytes + 4096));
    std::vector<Pattern> P;
    P.push_back({"A    fill", [&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, out, nch); }});
    P.push_back({"F256 dense front", [&] { hipLaunchKernelGGL((k_front<256>), dim3(256), dim3(256), 0, 0, out, nch); }});
    P.push_back({"R4-0 render pattern: chunks from the region start (16 waves/wg)", [&] { hipLaunchKernelGGL((k_wave_run<16, 0>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"R4-1 render pattern: 1 KiB-aligned stores per region", [&] { hipLaunchKernelGGL((k_wave_run<16, 1>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"R4-2 render pattern: 4 KiB-aligned bursts per region", [&] { hipLaunchKernelGGL((k_wave_run<16, 2>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"S-1  run as one stream, 1 KiB-aligned stores", [&] { hipLaunchKernelGGL((k_wave_stream<16, 0>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"S-4  run as one stream, 4 KiB-aligned bursts", [&] { hipLaunchKernelGGL((k_wave_stream<16, 1>), dim3(256), dim3(1024), 0, 0, out, nregions); }});
    P.push_back({"R4-0q 4 waves/wg x 1024 wgs: chunks from the region start", [&] { hipLaunchKernelGGL((k_wave_run<4, 0>), dim3(1024), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"R4-2q 4 waves/wg x 1024 wgs: 4 KiB-aligned bursts per region", [&] { hipLaunchKernelGGL((k_wave_run<4, 2>), dim3(1024), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"S-4q  4 waves/wg x 1024 wgs: one stream, 4 KiB-aligned bursts", [&] { hipLaunchKernelGGL((k_wave_stream<4, 1>), dim3(1024), dim3(256), 0, 0, out, nregions); }});
    P.push_back({"S-4w  4 waves/wg x 256 wgs: one stream, 4 KiB-aligned bursts", [&] { hipLaunchKernelGGL((k_wave_stream<4, 1>), dim3(256), dim3(256), 0, 0, out, nregions); }});
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run_for = [&](const Pattern& p, float seconds) -> float {
        double total = 0; long cnt = 0;
        p.launch(); CK(hipDeviceSynchronize());
        while (total < seconds * 1e3) {
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; i++) p.launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            total += ms; cnt += 20;
        }
        return (float)(total / cnt);
    };
    // correctness of the aligned variants' coverage: every chunk written exactly by construction is not checked
    // here (values differ); a coverage check on a small case:
    {
        CK(hipMemset(out, 0, bytes));
        hipLaunchKernelGGL((k_wave_run<16, 2>), dim3(256), dim3(1024), 0, 0, out, nregions);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> h(64 * 1024 * 4);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        size_t holes = 0;
        for (size_t i = 0; i < h.size() / 4; i++) if (h[4 * i + 3] != 4) holes++;
        CK(hipMemset(out, 0, bytes));
        hipLaunchKernelGGL((k_wave_stream<16, 1>), dim3(256), dim3(1024), 0, 0, out, nregions);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), (uint8_t*)out + bytes - h.size() * 4, h.size() * 4, hipMemcpyDeviceToHost));
        size_t holes2 = 0;
        for (size_t i = 0; i < h.size() / 4; i++) if (h[4 * i + 3] != 4) holes2++;
        printf("coverage check: R4-2 holes in the first MiB %zu, S-4 holes in the last MiB %zu\n", holes, holes2);
    }
    for (int r = 0; r < rounds; r++)
        for (auto& p : P) {
            const float ms = run_for(p, slice_s);
            printf("r%d %-70s %.4f ms %5.0f GB/s\n", r, p.name.c_str(), ms, bytes / ms / 1e6);
            fflush(stdout);
        }
    CK(hipFree(out));
    return 0;
}

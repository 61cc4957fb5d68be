// store_patterns4.hip — is the write-pattern gap a CHANNEL BALANCE effect?  Hypothesis after patterns 2/3:
// HBM channels interleave at ~4 KiB; a dense front (workgroup w writes block r*256 + w) puts every
// concurrently written block on its own channel, while per-wave / per-workgroup contiguous spans start at
// arbitrary strides and collide.  Test: contiguous spans whose length is an ODD vs EVEN number of 4 KiB
// blocks (writer i starts at block i * span: with an odd span, writers in lockstep sit on distinct
// channels), at wave and workgroup granularity; and the dense front with its base shifted.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns4.hip -o store_patterns4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr size_t TOTAL = (size_t)32768 * 28224;   // the bench workload's obs tensor, bytes (= 227556 blocks of 4 KiB)

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS;
    for (size_t c = (size_t)blockIdx.x * THREADS + threadIdx.x; c < nchunks; c += per_round) out[c] = make_uint4(1, 2, 3, 4);
}
// every WAVE writes its own contiguous span of `span_chunks` 16-byte chunks, 4 x 1 KiB stores back to back
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_wave_span(uint4* out, size_t nchunks, size_t span_chunks) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const size_t c0 = gw * span_chunks, c1 = min(nchunks, c0 + span_chunks);
    size_t c = c0 + lane;
    for (; c + 192 < c1; c += 256) { out[c] = make_uint4(1, 2, 3, 4); out[c + 64] = make_uint4(1, 2, 3, 4); out[c + 128] = make_uint4(1, 2, 3, 4); out[c + 192] = make_uint4(1, 2, 3, 4); }
    for (; c < c1; c += 64) out[c] = make_uint4(1, 2, 3, 4);
}
// every WORKGROUP writes its own contiguous span cooperatively (WPB KiB per step)
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_block_span(uint4* out, size_t nchunks, size_t span_chunks) {
    const size_t c0 = (size_t)blockIdx.x * span_chunks, c1 = min(nchunks, c0 + span_chunks);
    for (size_t c = c0 + threadIdx.x; c < c1; c += WPB * 64) out[c] = make_uint4(1, 2, 3, 4);
}
// every workgroup owns a contiguous span but its 4 waves write it as 4 KiB blocks: wave k takes blocks k, k+4, ...
// of the span (each wave: 4 consecutive 1 KiB stores = one aligned 4 KiB block)
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_block_span_wb(uint4* out, size_t nchunks, size_t span_chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t c0 = (size_t)blockIdx.x * span_chunks, c1 = min(nchunks, c0 + span_chunks);
    for (size_t b = c0 + (size_t)wave * 256; b < c1; b += (size_t)WPB * 256) {
        const size_t c = b + lane;
        if (c + 192 < c1) { out[c] = make_uint4(1, 2, 3, 4); out[c + 64] = make_uint4(1, 2, 3, 4); out[c + 128] = make_uint4(1, 2, 3, 4); out[c + 192] = make_uint4(1, 2, 3, 4); }
        else for (size_t cc = c; cc < c1 && cc < b + 256; cc += 64) out[cc] = make_uint4(1, 2, 3, 4);
    }
}

struct Pattern { std::string name; std::function<void()> launch; };

int main(int argc, char** argv) {
    const float slice_s = argc > 1 ? atof(argv[1]) : 0.25f;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    uint8_t* base; CK(hipMalloc(&base, TOTAL + (1 << 20)));
    const size_t nch = TOTAL / 16;
    std::vector<Pattern> P;
    auto at = [&](size_t off) { return reinterpret_cast<uint4*>(base + off); };
    P.push_back({"A     fill", [&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, at(0), nch); }});
    for (size_t off : {(size_t)0, (size_t)64, (size_t)1024, (size_t)2048}) {
        char nm[96]; snprintf(nm, 96, "F256  dense front, base + %zu B", off);
        P.push_back({nm, [&, off] { hipLaunchKernelGGL((k_front<256>), dim3(256), dim3(256), 0, 0, at(off), nch); }});
    }
    // wave spans: 4096 waves (16 per wg, 256 wgs) and 1024 waves (4 per wg, 256 wgs)
    for (size_t kib : {(size_t)220, (size_t)221, (size_t)222, (size_t)223, (size_t)224, (size_t)228, (size_t)225792 / 1024}) {
        char nm[96]; snprintf(nm, 96, "W16   4096 waves, span %zu KiB (%s x 4 KiB)", kib, (kib % 4) ? "frac" : ((kib / 4) % 2 ? "ODD" : "even"));
        const size_t span = kib * 64;
        const int wgs = (int)((nch + span * 16 - 1) / (span * 16));
        P.push_back({nm, [&, span, wgs] { hipLaunchKernelGGL((k_wave_span<16>), dim3(wgs), dim3(1024), 0, 0, at(0), nch, span); }});
    }
    {   // the render kernel's exact partition: 8 regions of 28224 B per wave (= 220.5 KiB)
        const size_t span = 8 * 28224 / 16;
        P.push_back({"W16   4096 waves, span 8 regions = 220.5 KiB (render)", [&, span] { hipLaunchKernelGGL((k_wave_span<16>), dim3(256), dim3(1024), 0, 0, at(0), nch, span); }});
    }
    for (size_t blocks : {(size_t)888, (size_t)889, (size_t)890, (size_t)891}) {
        char nm[96]; snprintf(nm, 96, "WG16  256 wgs x 16 waves, span %zu blocks (%s)", blocks, blocks % 2 ? "ODD" : "even");
        const size_t span = blocks * 256;
        P.push_back({nm, [&, span] { hipLaunchKernelGGL((k_block_span<16>), dim3((unsigned)((nch + span - 1) / span)), dim3(1024), 0, 0, at(0), nch, span); }});
    }
    for (size_t blocks : {(size_t)888, (size_t)889, (size_t)891}) {
        char nm[96]; snprintf(nm, 96, "WG4   256 wgs x 4 waves, span %zu blocks (%s)", blocks, blocks % 2 ? "ODD" : "even");
        const size_t span = blocks * 256;
        P.push_back({nm, [&, span] { hipLaunchKernelGGL((k_block_span<4>), dim3((unsigned)((nch + span - 1) / span)), dim3(256), 0, 0, at(0), nch, span); }});
        snprintf(nm, 96, "WG4b  256 wgs x 4 waves, wave = 4 KiB block, span %zu (%s)", blocks, blocks % 2 ? "ODD" : "even");
        P.push_back({nm, [&, span] { hipLaunchKernelGGL((k_block_span_wb<4>), dim3((unsigned)((nch + span - 1) / span)), dim3(256), 0, 0, at(0), nch, span); }});
    }
    for (size_t blocks : {(size_t)222, (size_t)223}) {
        char nm[96]; snprintf(nm, 96, "WG4   1024 wgs x 4 waves, span %zu blocks (%s)", blocks, blocks % 2 ? "ODD" : "even");
        const size_t span = blocks * 256;
        P.push_back({nm, [&, span] { hipLaunchKernelGGL((k_block_span<4>), dim3((unsigned)((nch + span - 1) / span)), dim3(256), 0, 0, at(0), nch, span); }});
    }
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
    for (int r = 0; r < rounds; r++)
        for (auto& p : P) {
            const float ms = run_for(p, slice_s);
            printf("r%d %-62s %.4f ms %5.0f GB/s\n", r, p.name.c_str(), ms, TOTAL / ms / 1e6);
            fflush(stdout);
        }
    CK(hipFree(base));
    return 0;
}

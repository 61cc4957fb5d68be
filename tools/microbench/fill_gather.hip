// fill_gather.hip — can a fill-shaped raster (one 16-byte chunk per thread, no persistent waves, no
// LDS) keep up when every chunk needs 2 x u16 tmap look-ups and 2 x 8-byte atlas gathers served by
// L1/L2?  Build: hipcc --offload-arch=gfx950 -O3 fill_gather.hip -o fill_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int VS = 7, TS = 8, TD = 6, PT = 3, PR = 21, P = 56, NAG = 3;
constexpr int IMG_CH = P * P * 3 / 16;          // 588 chunks per image
constexpr int ENV_CH = NAG * IMG_CH;            // 1764
constexpr int TMAP = NAG * VS * VS;             // 147 u16 per env

__global__ __launch_bounds__(256) void k_fill(uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_uint4(1, 2, 3, 4);
}

// one chunk per thread; tmap[e][147] (u16 dword offsets into atlas32) and atlas in global memory
template <int CHUNKS_PER_THREAD>
__global__ __launch_bounds__(256) void k_raster(uint4* __restrict__ out, const uint16_t* __restrict__ tmap,
                                                const uint32_t* __restrict__ atlas32, size_t nchunks) {
    size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < CHUNKS_PER_THREAD; u++) {
        size_t i = i0 + (size_t)u * gridDim.x * 256;
        if (i >= nchunks) return;
        uint32_t e = (uint32_t)(i / ENV_CH), c = (uint32_t)(i - (size_t)e * ENV_CH);
        uint32_t p0 = 2 * c, r = p0 / PR, pr = p0 - r * PR;
        uint32_t r1 = r, pr1 = pr + 1;
        if (pr1 == PR) { pr1 = 0; r1++; }
        const uint16_t* tm = tmap + (size_t)e * 160;
        uint32_t va = pr / PT, kp = pr - va * PT, vb = r >> 3, rr = r & 7;
        uint32_t a0 = tm[vb * VS + va] + rr * TD + kp * 2;
        va = pr1 / PT; kp = pr1 - va * PT; vb = r1 >> 3; rr = r1 & 7;
        uint32_t a1 = tm[vb * VS + va] + rr * TD + kp * 2;
        uint2 q0 = *reinterpret_cast<const uint2*>(atlas32 + a0);
        uint2 q1 = *reinterpret_cast<const uint2*>(atlas32 + a1);
        out[i] = make_uint4(q0.x, q0.y, q1.x, q1.y);
    }
}

template <typename F>
static float time_it(F launch, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int B = 32768;
    const size_t nch = (size_t)B * ENV_CH, bytes = nch * 16;
    uint4* out; CK(hipMalloc(&out, bytes));
    const int NT = 28;                                  // tiles
    std::vector<uint32_t> atlas(4 * NT * 48 + 16, 0x01020304u);
    std::vector<uint16_t> tmap((size_t)B * 160);
    srand(1);
    for (size_t i = 0; i < tmap.size(); i++) {
        int t = rand() % 100; int tile = t < 45 ? 0 : t < 80 ? 1 : t < 92 ? 2 : 3 + rand() % 24;   // mostly shadow/empty/wall
        int o = (int)((i / 160) % 4);
        tmap[i] = (uint16_t)((o * NT + tile) * 48);
    }
    uint32_t* d_atlas; uint16_t* d_tmap;
    CK(hipMalloc(&d_atlas, atlas.size() * 4)); CK(hipMemcpy(d_atlas, atlas.data(), atlas.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tmap, tmap.size() * 2)); CK(hipMemcpy(d_tmap, tmap.data(), tmap.size() * 2, hipMemcpyHostToDevice));
    auto rep = [&](const char* nm, float ms) { printf("%-50s %.4f ms  %.0f GB/s\n", nm, ms, bytes / ms / 1e6); };
    for (int r = 0; r < 2; r++) {
        rep("fill", time_it([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        rep("raster, 1 chunk/thread, tmap+atlas via L1/L2", time_it([&] { hipLaunchKernelGGL((k_raster<1>), dim3((nch + 255) / 256), dim3(256), 0, 0, out, d_tmap, d_atlas, nch); }, 20));
        rep("raster, 4 chunks/thread (grid-strided)", time_it([&] { hipLaunchKernelGGL((k_raster<4>), dim3((nch / 4 + 255) / 256), dim3(256), 0, 0, out, d_tmap, d_atlas, nch); }, 20));
    }
    return 0;
}

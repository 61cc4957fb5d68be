// store_patterns3.hip — follow-up to store_patterns2: (1) is the gap between "wave per region" and a
// dense front a TAIL effect (static partition, waves finish unevenly) or a steady-state pattern effect?
// -> every pattern at 32768 and at 262144 regions; (2) the store side of a producer / streamer raster:
// one workgroup per CU whose 4 streamer waves write one 28 KB region per round cooperatively (dense
// front at region granularity across the chip), alone and next to 12 waves doing LDS work; streamers
// fed from an LDS ring that producer waves fill (flags in LDS); (3) store flavours (plain / nt / sc1)
// and what they cost at the next kernel boundary.
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns3.hip -o store_patterns3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 28224;
constexpr int RCH = REGION / 16;   // 1764 chunks of 16 B
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int FLAVOR>   // 0 plain, 1 nontemporal, 2 sc1 (write-through, line dropped from L2), 3 sc0 sc1
__device__ __forceinline__ void st16(uint4* p, uint4 v) {
    if constexpr (FLAVOR == 0) *p = v;
    else if constexpr (FLAVOR == 1) { u32x4 nv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4*>(p)); }
    else if constexpr (FLAVOR == 2) { u32x4 nv = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(nv) : "memory"); }
    else { u32x4 nv = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(nv) : "memory"); }
}

__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
__global__ void k_tiny(uint32_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

template <int THREADS, int FLAVOR>
__global__ __launch_bounds__(THREADS) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS;
    for (size_t c = (size_t)blockIdx.x * THREADS + threadIdx.x; c < nchunks; c += per_round) st16<FLAVOR>(out + c, make_uint4(1, 2, 3, 4));
}
// render pattern: wave per region, contiguous run per wave, x4 stores
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_wave_run(uint4* out, int nregions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = (nregions + gridDim.x * WPB - 1) / (gridDim.x * WPB);
    const int e0 = (blockIdx.x * WPB + wave) * per_wave, e1 = min(nregions, e0 + per_wave);
    for (int e = e0; e < e1; e++) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + 192 < RCH; c += 256) { o[c] = make_uint4(e, c, 0, 4); o[c + 64] = make_uint4(e, c, 1, 4); o[c + 128] = make_uint4(e, c, 2, 4); o[c + 192] = make_uint4(e, c, 3, 4); }
        for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// wave per region, regions handed out dynamically from one global counter, BATCH regions per grab
template <int WPB, int BATCH>
__global__ __launch_bounds__(WPB * 64) void k_wave_dyn(uint4* out, int nregions, int* counter) {
    const int lane = threadIdx.x & 63;
    while (true) {
        int e0 = 0;
        if (lane == 0) e0 = atomicAdd(counter, BATCH);
        e0 = __shfl(e0, 0);
        if (e0 >= nregions) break;
        const int e1 = min(nregions, e0 + BATCH);
        for (int e = e0; e < e1; e++) {
            uint4* o = out + (size_t)e * RCH;
            int c = lane;
            for (; c + 192 < RCH; c += 256) { o[c] = make_uint4(e, c, 0, 4); o[c + 64] = make_uint4(e, c, 1, 4); o[c + 128] = make_uint4(e, c, 2, 4); o[c + 192] = make_uint4(e, c, 3, 4); }
            for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
        }
    }
}
// STREAMER TEAM: the first SW waves of the workgroup write region (q * gridDim + vb) in round q cooperatively;
// MAP 0: vb = blockIdx (adjacent CUs / alternating XCDs write adjacent regions), 1: XCD-contiguous spans.
// The other waves of the workgroup (if any) do LDS busy work until the streamers are done.
template <int WPB, int SW, int MAP, int FLAVOR, int UNR>
__global__ __launch_bounds__(WPB * 64) void k_team(uint4* out, int nregions) {
    __shared__ uint32_t busy[WPB][64];
    __shared__ volatile int done_flag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) done_flag = 0;
    __syncthreads();
    const int nb = gridDim.x;
    int vb = blockIdx.x;
    if (MAP == 1) { const int per = (nb + 7) / 8; vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    if (wave < SW) {
        const int t = threadIdx.x;   // 0 .. SW*64-1
        for (int e = vb; e < nregions; e += nb) {
            uint4* o = out + (size_t)e * RCH;
            int c = t;
            for (; c + (UNR - 1) * SW * 64 < RCH; c += UNR * SW * 64)
#pragma unroll
                for (int u = 0; u < UNR; u++) st16<FLAVOR>(o + c + u * SW * 64, make_uint4(e, c, u, 4));
            for (; c < RCH; c += SW * 64) st16<FLAVOR>(o + c, make_uint4(e, c, 9, 4));
        }
        if (WPB > SW) {
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0) atomicAdd((int*)&done_flag, 1);
        }
    } else {
        uint32_t x = lane * 2654435761u;
        while (done_flag < SW) {
#pragma unroll 8
            for (int i = 0; i < 64; i++) { busy[wave][(lane + i) & 63] = x; x = x * 1664525u + busy[wave][(lane * 7 + i) & 63]; }
        }
        if (x == 0x12345) out[0] = make_uint4(x, 0, 0, 0);
    }
}
// PRODUCER / STREAMER through an LDS ring: producers assemble a region's 28224 bytes into slot (q % SLOTS)
// as 1176 segments of 24 B (3 x ds_write_b64 per lane, like the raster's tile-row copies) and publish it;
// the 4 streamer waves read it linearly (ds_read_b128) and store it.  One workgroup per CU.
template <int NP, int SLOTS>
__global__ __launch_bounds__((4 + NP) * 64) void k_ring(uint4* out, int nregions) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    volatile int* ready = reinterpret_cast<volatile int*>(smem);          // [SLOTS] region seq published (q + 1)
    volatile int* freed = ready + SLOTS;                                   // [SLOTS] region seq consumed (q + 1), per streamer wave count
    int* claims = const_cast<int*>(freed + SLOTS);                         // next q to claim by a producer
    uint8_t* ring = smem + 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < SLOTS) { ready[threadIdx.x] = 0; freed[threadIdx.x] = 0; }
    if (threadIdx.x == 0) *claims = 0;
    __syncthreads();
    const int nb = gridDim.x;
    const int nq = (nregions - (int)blockIdx.x + nb - 1) / nb;             // regions of this workgroup
    if (wave < 4) {
        const int t = threadIdx.x;
        for (int q = 0; q < nq; q++) {
            const int slot = q % SLOTS;
            while (ready[slot] < q + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint4* src = reinterpret_cast<const uint4*>(ring + (size_t)slot * 28672);
            uint4* o = out + (size_t)(blockIdx.x + (size_t)q * nb) * RCH;
            int c = t;
            for (; c + 768 < RCH; c += 1024) {
                const uint4 v0 = src[c], v1 = src[c + 256], v2 = src[c + 512], v3 = src[c + 768];
                o[c] = v0; o[c + 256] = v1; o[c + 512] = v2; o[c + 768] = v3;
            }
            for (; c < RCH; c += 256) o[c] = src[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS reads of this slot are done
            if (lane == 0) atomicAdd((int*)&freed[slot], 1);           // 4 streamer waves -> +4 per use
        }
    } else {
        while (true) {
            int q = 0;
            if (lane == 0) q = atomicAdd(claims, 1);
            q = __shfl(q, 0);
            if (q >= nq) break;
            const int slot = q % SLOTS, use = q / SLOTS;
            while (freed[slot] < 4 * use) __builtin_amdgcn_s_sleep(1);   // the previous occupant was streamed out
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint8_t* dst = ring + (size_t)slot * 28672;
            for (int g = lane; g < 1176; g += 64) {
                uint64_t* d = reinterpret_cast<uint64_t*>(dst + g * 24);
                const uint64_t v = (uint64_t)q * 1315423911ull + g;
                d[0] = v; d[1] = v + 1; d[2] = v + 2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[slot] = q + 1;
            // NOTE: producers publish out of order only across different slots; slot order is kept by `use`
        }
    }
}

struct Pattern { std::string name; std::function<void(int)> launch; };

int main(int argc, char** argv) {
    const float slice_s = argc > 1 ? atof(argv[1]) : 0.3f;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const int NBIG = 262144, NSMALL = 32768;
    const size_t bytes_big = (size_t)NBIG * REGION;
    uint4* out; CK(hipMalloc(&out, bytes_big));
    int* counter; CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
    uint32_t* tiny; CK(hipMalloc(&tiny, 64)); CK(hipMemset(tiny, 0, 64));
    std::vector<Pattern> P;
    auto nch = [](int n) { return (size_t)n * RCH; };
    P.push_back({"A    fill", [&](int n) { hipLaunchKernelGGL(k_fill, dim3((nch(n) + 255) / 256), dim3(256), 0, 0, out, nch(n)); }});
    P.push_back({"F256 dense front 256thr x 256wg plain", [&](int n) { hipLaunchKernelGGL((k_front<256, 0>), dim3(256), dim3(256), 0, 0, out, nch(n)); }});
    P.push_back({"F256 dense front nt", [&](int n) { hipLaunchKernelGGL((k_front<256, 1>), dim3(256), dim3(256), 0, 0, out, nch(n)); }});
    P.push_back({"F256 dense front sc1", [&](int n) { hipLaunchKernelGGL((k_front<256, 2>), dim3(256), dim3(256), 0, 0, out, nch(n)); }});
    P.push_back({"F256 dense front sc0 sc1", [&](int n) { hipLaunchKernelGGL((k_front<256, 3>), dim3(256), dim3(256), 0, 0, out, nch(n)); }});
    P.push_back({"R4   16-wave wg, wave/region run x4, 256 wgs (render)", [&](int n) { hipLaunchKernelGGL((k_wave_run<16>), dim3(256), dim3(1024), 0, 0, out, n); }});
    P.push_back({"R4q  4-wave wg, wave/region run x4, 256 wgs", [&](int n) { hipLaunchKernelGGL((k_wave_run<4>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"D1   16-wave wg, wave/region DYNAMIC 1/grab", [&](int n) { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL((k_wave_dyn<16, 1>), dim3(256), dim3(1024), 0, 0, out, n, counter); }});
    P.push_back({"D4   16-wave wg, wave/region DYNAMIC 4/grab", [&](int n) { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL((k_wave_dyn<16, 4>), dim3(256), dim3(1024), 0, 0, out, n, counter); }});
    P.push_back({"D1q  4-wave wg x 256, wave/region DYNAMIC 1/grab", [&](int n) { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL((k_wave_dyn<4, 1>), dim3(256), dim3(256), 0, 0, out, n, counter); }});
    P.push_back({"T4i  4 streamers/CU, region/round, adjacent CUs", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 0, 1>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4i2 same, x2 unrolled", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 0, 2>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4i4 same, x4 unrolled", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 0, 4>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4x  4 streamers/CU, XCD-contiguous spans", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 1, 0, 1>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4n  4 streamers/CU, nt stores", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 1, 1>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4s  4 streamers/CU, sc1 stores", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 2, 1>), dim3(256), dim3(256), 0, 0, out, n); }});
    P.push_back({"T8i  8 streamers/CU", [&](int n) { hipLaunchKernelGGL((k_team<8, 8, 0, 0, 1>), dim3(256), dim3(512), 0, 0, out, n); }});
    P.push_back({"T2i  2 streamers/CU", [&](int n) { hipLaunchKernelGGL((k_team<2, 2, 0, 0, 1>), dim3(256), dim3(128), 0, 0, out, n); }});
    P.push_back({"T4w  2 wgs/CU x 4 streamers", [&](int n) { hipLaunchKernelGGL((k_team<4, 4, 0, 0, 1>), dim3(512), dim3(256), 0, 0, out, n); }});
    P.push_back({"T4b  4 streamers + 12 LDS-busy waves /CU", [&](int n) { hipLaunchKernelGGL((k_team<16, 4, 0, 0, 1>), dim3(256), dim3(1024), 0, 0, out, n); }});
    P.push_back({"T4b8 4 streamers + 4 LDS-busy waves /CU", [&](int n) { hipLaunchKernelGGL((k_team<8, 4, 0, 0, 1>), dim3(256), dim3(512), 0, 0, out, n); }});
    {
        const size_t lds3 = 256 + 3 * 28672, lds4 = 256 + 4 * 28672;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<12, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<12, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        P.push_back({"RG12/3 ring: 12 producers, 3 slots, 4 streamers", [=, &out](int n) { hipLaunchKernelGGL((k_ring<12, 3>), dim3(256), dim3(1024), lds3, 0, out, n); }});
        P.push_back({"RG4/3  ring: 4 producers, 3 slots, 4 streamers", [=, &out](int n) { hipLaunchKernelGGL((k_ring<4, 3>), dim3(256), dim3(512), lds3, 0, out, n); }});
        P.push_back({"RG12/4 ring: 12 producers, 4 slots, 4 streamers", [=, &out](int n) { hipLaunchKernelGGL((k_ring<12, 4>), dim3(256), dim3(1024), lds4, 0, out, n); }});
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run_for = [&](const Pattern& p, int n, float seconds, bool with_tiny) -> float {
        double total = 0; long cnt = 0;
        p.launch(n); CK(hipDeviceSynchronize());
        while (total < seconds * 1e3) {
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; i++) { p.launch(n); if (with_tiny) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, tiny); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            total += ms; cnt += 20;
        }
        return (float)(total / cnt);
    };
    for (int r = 0; r < rounds; r++)
        for (auto& p : P) {
            const float ms_s = run_for(p, NSMALL, slice_s, false), ms_b = run_for(p, NBIG, slice_s, false);
            const float ms_t = run_for(p, NSMALL, slice_s * 0.5f, true);
            const double bs = (double)NSMALL * REGION, bb = (double)NBIG * REGION;
            // fixed + slope model from the two sizes
            const double slope = (ms_b - ms_s) / (NBIG - NSMALL), fixed = ms_s - slope * NSMALL;
            printf("r%d %-52s 32k: %.4f ms %5.0f GB/s | 256k: %.4f ms %5.0f GB/s | steady %5.0f GB/s fixed %5.1f us | +tiny kernel: %.4f ms (boundary %+5.1f us)\n",
                   r, p.name.c_str(), ms_s, bs / ms_s / 1e6, ms_b, bb / ms_b / 1e6, REGION / slope / 1e6, fixed * 1e3, ms_t, (ms_t - ms_s) * 1e3);
            fflush(stdout);
        }
    CK(hipFree(out));
    return 0;
}

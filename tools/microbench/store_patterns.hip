// store_patterns.hip — which HBM *write* pattern gets closest to the fill ceiling on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns ; run on the GPU box.
// All variants write the same 32768 x 28224 B (the bench workload's obs tensor).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 28224;            // bytes per env (3 x 56 x 56 x 3)
constexpr int RCH = REGION / 16;         // 1764 chunks

// A: fill-like, one 16-B store per thread
__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}
// B: persistent, wave per region (the render kernel's pattern)
__global__ __launch_bounds__(256) void k_wave_region(uint4* out, int nregions) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * 4 + wave; e < nregions; e += gridDim.x * 4) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// C: persistent, workgroup (4 waves) per region: 4 KiB contiguous per step
__global__ __launch_bounds__(256) void k_block_region(uint4* out, int nregions) {
    for (int e = blockIdx.x; e < nregions; e += gridDim.x) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = threadIdx.x; c < RCH; c += 256) o[c] = make_uint4(e, c, 3, 4);
    }
}
// D: persistent, workgroup per PAIR of regions interleaved by wave pairs (2 KiB runs)
__global__ __launch_bounds__(256) void k_halfblock_region(uint4* out, int nregions) {
    int half = threadIdx.x >> 7, t = threadIdx.x & 127;
    for (int e = blockIdx.x * 2 + half; e < nregions; e += gridDim.x * 2) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = t; c < RCH; c += 128) o[c] = make_uint4(e, c, 3, 4);
    }
}
// E: wave per region but a wave issues 4 stores back-to-back per loop trip (deeper store queue)
__global__ __launch_bounds__(256) void k_wave_region_x4(uint4* out, int nregions) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * 4 + wave; e < nregions; e += gridDim.x * 4) {
        uint4* o = out + (size_t)e * RCH;
        int c = lane;
        for (; c + 192 < RCH; c += 256) {
            o[c] = make_uint4(e, c, 3, 4); o[c + 64] = make_uint4(e, c, 3, 4);
            o[c + 128] = make_uint4(e, c, 3, 4); o[c + 192] = make_uint4(e, c, 3, 4);
        }
        for (; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// F: non-persistent: one wave per region, grid = nregions/4 workgroups
__global__ __launch_bounds__(256) void k_wave_region_np(uint4* out, int nregions) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int e = blockIdx.x * 4 + wave;
    if (e >= nregions) return;
    uint4* o = out + (size_t)e * RCH;
    for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
}
// G: wave per region, region order bit-swizzled so that concurrently running waves of one CU write
// neighbouring regions (blockIdx -> XCD is b % 8: give each XCD a contiguous span)
__global__ __launch_bounds__(256) void k_wave_region_xcd(uint4* out, int nregions) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int nb = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int per_xcd = (nb + 7) / 8;
    int vb = xcd * per_xcd + idx;           // virtual block id: contiguous per XCD
    for (int e = vb * 4 + wave; e < nregions; e += nb * 4) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}

// H: pattern B + what the render kernel does per env: a register prefetch of the NEXT region's small
// input (one dword per lane) that is consumed (written to LDS) at the top of the next trip.  On
// CDNA the wait for that load is a vmcnt wait, and vmcnt also counts this wave's stores.
__global__ __launch_bounds__(256) void k_wave_region_vmem_in(uint4* out, const uint32_t* in, int nregions) {
    __shared__ uint32_t sm[4][64];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int e = blockIdx.x * 4 + wave, es = gridDim.x * 4;
    uint32_t pf = e < nregions ? in[(size_t)e * 64 + lane] : 0;
    for (; e < nregions; e += es) {
        sm[wave][lane] = pf;
        if (e + es < nregions) pf = in[(size_t)(e + es) * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t x = sm[wave][(lane + 1) & 63];
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, x, 4);
    }
}
// I: same input, fetched through the scalar cache (lgkmcnt domain: independent of the stores)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_wave_region_smem_in(uint4* out, const uint32_t* in, int nregions) {
    __shared__ uint32_t sm[4][64];
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int e = blockIdx.x * 4 + wave, es = gridDim.x * 4;
    typedef const __attribute__((address_space(4))) u32x16* cp16;
    for (; e < nregions; e += es) {
        cp16 src = (cp16)(in + (size_t)e * 64);
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32x16 s16 = src[q];
#pragma unroll
            for (int i = 0; i < 16; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
                uint32_t sv = s16[i];
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sv), "n"(q * 16 + i));
#endif
            }
        }
        sm[wave][lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t x = sm[wave][(lane + 1) & 63];
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, x, 4);
    }
}
// J: H without prefetch (load at the top, consumed at once)
__global__ __launch_bounds__(256) void k_wave_region_vmem_nopf(uint4* out, const uint32_t* in, int nregions) {
    __shared__ uint32_t sm[4][64];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * 4 + wave; e < nregions; e += gridDim.x * 4) {
        sm[wave][lane] = in[(size_t)e * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t x = sm[wave][(lane + 1) & 63];
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, x, 4);
    }
}

// K: 16-wave workgroup per region (all 1024 threads stream one region, then the next)
__global__ __launch_bounds__(1024) void k_bigblock_region(uint4* out, int nregions) {
    for (int e = blockIdx.x; e < nregions; e += gridDim.x) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = threadIdx.x; c < RCH; c += 1024) o[c] = make_uint4(e, c, 3, 4);
    }
}
// L: 16-wave workgroup, wave per region, 16 adjacent regions per workgroup trip (the render kernel now)
__global__ __launch_bounds__(1024) void k_wave_region_16(uint4* out, int nregions) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * 16 + wave; e < nregions; e += gridDim.x * 16) {
        uint4* o = out + (size_t)e * RCH;
        for (int c = lane; c < RCH; c += 64) o[c] = make_uint4(e, c, 3, 4);
    }
}
// M: 16-wave workgroup walks 16 adjacent regions cooperatively, one after the other
__global__ __launch_bounds__(1024) void k_bigblock_16seq(uint4* out, int nregions) {
    for (int e0 = blockIdx.x * 16; e0 < nregions; e0 += gridDim.x * 16) {
        uint4* o = out + (size_t)e0 * RCH;
        int tot = RCH * min(16, nregions - e0);
        for (int c = threadIdx.x; c < tot; c += 1024) o[c] = make_uint4(e0, c, 3, 4);
    }
}

// N: non-persistent, NS consecutive stores per thread (wave writes NS consecutive KiB): between fill
// (NS = 1) and "wave per region" (NS = 28)
template <int NS>
__global__ __launch_bounds__(256) void k_fill_n(uint4* out, size_t nchunks) {
    size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    size_t base = wave * 64 * NS + lane;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        size_t c = base + (size_t)i * 64;
        if (c < nchunks) out[c] = make_uint4(1, 2, 3, (unsigned)i);
    }
}
// P: persistent workgroups, dense moving front: in round r workgroup w writes the contiguous block
// (r * gridDim + w) of BLK bytes (= what fill does, but with long-lived waves)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_front(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS;
    for (size_t c = (size_t)blockIdx.x * THREADS + threadIdx.x; c < nchunks; c += per_round) out[c] = make_uint4(1, 2, 3, 4);
}
// Q: like P but each workgroup writes RUN consecutive blocks per round (longer runs per workgroup)
template <int THREADS, int RUN>
__global__ __launch_bounds__(THREADS) void k_front_run(uint4* out, size_t nchunks) {
    const size_t per_round = (size_t)gridDim.x * THREADS * RUN;
    for (size_t c0 = (size_t)blockIdx.x * THREADS * RUN; c0 < nchunks; c0 += per_round)
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            size_t c = c0 + (size_t)i * THREADS + threadIdx.x;
            if (c < nchunks) out[c] = make_uint4(1, 2, 3, 4);
        }
}

// R: P (persistent dense front, 4 KiB blocks) + what a real raster needs per chunk: the block's env
// tmap fetched from global into LDS (prefetched one round ahead), then per chunk 2 x (u16 tmap read
// -> 8-byte atlas read) from LDS.  UNR = chunks per thread per round handled with independent chains
// (UNR blocks of 4 KiB apart by gridDim*4KiB, i.e. still one store per wave per dense front).
template <int UNR>
__global__ __launch_bounds__(256) void k_front_lds(uint4* out, const uint16_t* __restrict__ tmaps, size_t nchunks) {
    __shared__ uint32_t atlas[4 * 28 * 48];
    __shared__ uint16_t tm[2][UNR][2 * 160];
    for (int i = threadIdx.x; i < 4 * 28 * 48; i += 256) atlas[i] = i * 2654435761u;
    const size_t per_round = (size_t)gridDim.x * 256;
    auto load_tm = [&](int buf, size_t round_c0) {
        for (int u = 0; u < UNR; u++) {
            size_t c0 = round_c0 + (size_t)u * per_round;
            size_t e0 = c0 / 1764;
            if (threadIdx.x < 160 && c0 < nchunks) {
                tm[buf][u][threadIdx.x] = tmaps[e0 * 160 + threadIdx.x];
                tm[buf][u][160 + threadIdx.x] = tmaps[(e0 + 1) * 160 + threadIdx.x];
            }
        }
    };
    size_t c0 = (size_t)blockIdx.x * 256;
    load_tm(0, c0);
    __syncthreads();
    int buf = 0;
    for (; c0 < nchunks; c0 += per_round * UNR) {
        if (c0 + per_round * UNR < nchunks) load_tm(buf ^ 1, c0 + per_round * UNR);
        uint4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            size_t cb = c0 + (size_t)u * per_round, c = cb + threadIdx.x;
            size_t e0 = cb / 1764;
            uint32_t ci = (uint32_t)(c - e0 * 1764);             // may run into env e0+1 (second tmap half)
            uint32_t sel = ci >= 1764 ? 160u : 0u;
            if (ci >= 1764) ci -= 1764;
            uint32_t p0 = 2 * ci, r = p0 / 21, pr = p0 - r * 21, r1 = r, pr1 = pr + 1;
            if (pr1 == 21) { pr1 = 0; r1++; }
            uint32_t va = pr / 3, kp = pr - va * 3, vb = r >> 3, rr = r & 7;
            uint32_t a0 = (tm[buf][u][sel + (vb * 7 + va) % 147] % (4 * 28)) * 48 + rr * 6 + kp * 2;
            va = pr1 / 3; kp = pr1 - va * 3; vb = r1 >> 3; rr = r1 & 7;
            uint32_t a1 = (tm[buf][u][sel + (vb * 7 + va) % 147] % (4 * 28)) * 48 + rr * 6 + kp * 2;
            uint2 q0 = *reinterpret_cast<const uint2*>(&atlas[a0]);
            uint2 q1 = *reinterpret_cast<const uint2*>(&atlas[a1]);
            v[u] = make_uint4(q0.x, q0.y, q1.x, q1.y);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            size_t c = c0 + (size_t)u * per_round + threadIdx.x;
            if (c < nchunks) out[c] = v[u];
        }
        __syncthreads();
        buf ^= 1;
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
    const int nregions = 32768;
    const size_t bytes = (size_t)nregions * REGION;
    uint4* out; CK(hipMalloc(&out, bytes));
    uint32_t* in; CK(hipMalloc(&in, (size_t)nregions * 256 + 4096)); CK(hipMemset(in, 1, (size_t)nregions * 256 + 4096));
    uint16_t* tmaps; CK(hipMalloc(&tmaps, (size_t)(nregions + 2) * 320)); CK(hipMemset(tmaps, 3, (size_t)(nregions + 2) * 320));
    const size_t nch = bytes / 16;
    auto report = [&](const char* name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int rep = 0; rep < 2; rep++) {
        report("A fill (1 store/thread)", time_it([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        for (int per_cu : {5}) {
            char nm[96];
            int blocks = 256 * per_cu;
            snprintf(nm, 96, "B wave/region persistent, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region, dim3(blocks), dim3(256), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "C wg/region persistent, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_block_region, dim3(blocks), dim3(256), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "D half-wg/region persistent, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_halfblock_region, dim3(blocks), dim3(256), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "E wave/region x4 unrolled, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_x4, dim3(blocks), dim3(256), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "H wave/region + VMEM prefetched input, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_vmem_in, dim3(blocks), dim3(256), 0, 0, out, in, nregions); }, 20));
            snprintf(nm, 96, "I wave/region + SMEM input, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_smem_in, dim3(blocks), dim3(256), 0, 0, out, in, nregions); }, 20));
            snprintf(nm, 96, "J wave/region + VMEM input no prefetch, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_vmem_nopf, dim3(blocks), dim3(256), 0, 0, out, in, nregions); }, 20));
            snprintf(nm, 96, "G wave/region XCD-contiguous, %d wg/CU", per_cu);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_xcd, dim3(blocks), dim3(256), 0, 0, out, nregions); }, 20));
        }
        for (int nb : {256}) {
            char nm[96];
            snprintf(nm, 96, "K 16-wave wg/region, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_bigblock_region, dim3(nb), dim3(1024), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "L 16-wave wg, wave/region, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_wave_region_16, dim3(nb), dim3(1024), 0, 0, out, nregions); }, 20));
            snprintf(nm, 96, "M 16-wave wg, 16 regions cooperatively, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_bigblock_16seq, dim3(nb), dim3(1024), 0, 0, out, nregions); }, 20));
        }
        for (int nb : {256, 384, 512}) {
            char nm[96];
            snprintf(nm, 96, "R dense front + tmap/atlas LDS look-ups x1, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front_lds<1>), dim3(nb), dim3(256), 0, 0, out, tmaps, nch); }, 20));
            snprintf(nm, 96, "R dense front + tmap/atlas LDS look-ups x2, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front_lds<2>), dim3(nb), dim3(256), 0, 0, out, tmaps, nch); }, 20));
            snprintf(nm, 96, "R dense front + tmap/atlas LDS look-ups x4, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front_lds<4>), dim3(nb), dim3(256), 0, 0, out, tmaps, nch); }, 20));
        }
        report("N fill, 2 stores/thread", time_it([&] { hipLaunchKernelGGL((k_fill_n<2>), dim3((nch / 2 + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        report("N fill, 4 stores/thread", time_it([&] { hipLaunchKernelGGL((k_fill_n<4>), dim3((nch / 4 + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        report("N fill, 8 stores/thread", time_it([&] { hipLaunchKernelGGL((k_fill_n<8>), dim3((nch / 8 + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        report("N fill, 28 stores/thread", time_it([&] { hipLaunchKernelGGL((k_fill_n<28>), dim3((nch / 28 + 255) / 256), dim3(256), 0, 0, out, nch); }, 20));
        for (int nb : {256, 512, 1024, 2048}) {
            char nm[96];
            snprintf(nm, 96, "P persistent dense front, 256 thr, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front<256>), dim3(nb), dim3(256), 0, 0, out, nch); }, 20));
            snprintf(nm, 96, "P persistent dense front, 1024 thr, %d wgs", nb / 4);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front<1024>), dim3(nb / 4), dim3(1024), 0, 0, out, nch); }, 20));
            snprintf(nm, 96, "Q dense front, 256 thr x 4-block runs, %d wgs", nb);
            report(nm, time_it([&] { hipLaunchKernelGGL((k_front_run<256, 4>), dim3(nb), dim3(256), 0, 0, out, nch); }, 20));
        }
        report("F wave/region non-persistent", time_it([&] { hipLaunchKernelGGL(k_wave_region_np, dim3(nregions / 4), dim3(256), 0, 0, out, nregions); }, 20));
    }
    CK(hipFree(out));
    return 0;
}

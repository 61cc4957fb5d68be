// Does a DS instruction issued behind an LDS-DMA load (global_load_lds) wait for the load?  (gfx950)
// Every wave: t0; one load of 16 bytes per lane from a cold address — mode 0: global_load_lds_dwordx4 into the wave's
// landing zone, mode 1: a plain global_load_dwordx4 into registers, mode 2: no load —; t1; a ds_write + ds_read of an
// UNRELATED LDS word + s_waitcnt lgkmcnt(0); t2; s_waitcnt vmcnt(0); t3.  Prints the means of t2 - t1 and t3 - t1.
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_dma_stall.hip -o /tmp/lds_dma_stall && /tmp/lds_dma_stall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, size_t stride, unsigned long long* out, int mode, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* zone = reinterpret_cast<uint32_t*>(smem) + wave * 512;          // 2 KiB per wave: [0, 1 KiB) landing zone, then scratch
    uint32_t* mine = zone + 256 + lane;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    const uint4* p = src + gw * stride + lane;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    uint4 v = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = now();
    if (mode == 0) __builtin_amdgcn_global_load_lds((gptr)p, (lptr)zone, 16, 0, 0);
    else if (mode == 1) v = *p;
    const unsigned long long t1 = now();
    const uint32_t laddr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)mine;
    uint32_t r;
    asm volatile("ds_write_b32 %1, %2\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(laddr), "v"(lane) : "memory");
    const unsigned long long t2 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = now();
    if (mode == 1) r += v.x + v.y + v.z + v.w;
    if (mode == 0) r += zone[lane];
    if (r == 0xdeadbeefu) *sink = r;
    if (lane == 0) { out[gw * 4 + 0] = t0; out[gw * 4 + 1] = t1; out[gw * 4 + 2] = t2; out[gw * 4 + 3] = t3; }
}

int main() {
    const int blocks = 1024, waves = blocks * 4;
    const size_t stride = 4096;          // uint4 between waves: 64 KiB apart, cold lines
    uint4* src;
    unsigned long long* out;
    uint32_t* sink;
    (void)hipMalloc(&src, (size_t)waves * stride * sizeof(uint4));
    (void)hipMemset(src, 1, (size_t)waves * stride * sizeof(uint4));
    (void)hipMalloc(&out, waves * 4 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 4);
    uint8_t* trash;
    (void)hipMalloc(&trash, 1u << 30);
    std::vector<unsigned long long> h(waves * 4);
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 3; rep++) {
            (void)hipMemset(trash, rep, 1u << 30);         // the loads miss every cache
            (void)hipDeviceSynchronize();
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 8192, 0, src, stride, out, mode, sink);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double a = 0, b = 0, c = 0;
            for (int w = 0; w < waves; w++) { a += h[w * 4 + 1] - h[w * 4]; b += h[w * 4 + 2] - h[w * 4 + 1]; c += h[w * 4 + 3] - h[w * 4 + 1]; }
            printf("mode %d (%s) rep %d: issue %.2f us;  DS write+read behind it, lgkmcnt(0): %.2f us;  vmcnt(0): %.2f us after the load's issue\n",
                   mode, mode == 0 ? "global_load_lds" : mode == 1 ? "global_load to registers" : "no load", rep, a / waves / 100.0, b / waves / 100.0, c / waves / 100.0);
        }
    return 0;
}

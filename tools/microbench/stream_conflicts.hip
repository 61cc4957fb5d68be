// stream_conflicts.hip — which sets of CONCURRENT write streams does the HBM of an MI355X absorb well?
//
// Round 4, visit 2 (profiles/r04/scan_arena_offsets.txt): the obs raster's speed is a smooth function of WHERE in
// one 40 GiB allocation its 882 MiB buffer lies, with one valley — 0.153 ms against 0.193 — centred exactly where
// the buffer straddles the allocation's 32 GiB | 8 GiB physical block boundary half and half; the TLB counters see
// ~75 misses per launch and the L2's write-request counters are identical for fast and slow buffers: the class is
// decided behind the L2, in how the memory controllers' banks take 4 096 concurrent sequential streams whose start
// addresses are 441 x 512 B apart (streams w and w + 2^k share their low 9 + k address bits).
//
// This program writes that pattern — W waves, each streaming its own `run` bytes in 4 KiB bursts of 1 KiB
// wave-stores, all waves resident — from a table of per-wave start offsets and rotations, into physically
// contiguous memory (the first 30 GiB of one 40 GiB allocation), and sweeps: an extra gap in the middle of the
// stream set (what the block boundary does), the stride between streams, per-wave rotations of the run (what the
// raster could do without changing the output layout), and the number of streams.
// Build: hipcc --offload-arch=gfx950 -O3 stream_conflicts.hip -o stream_conflicts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// wave w: `run` bytes (a multiple of 16) starting at base + start[w]; it begins rot[w] bytes (a multiple of 16) into
// its run and wraps around.  4 x 1 KiB stores per trip, like the raster's chunk loop.
__global__ __launch_bounds__(1024) void k_streams(uint8_t* base, const uint64_t* __restrict__ start, const uint32_t* __restrict__ rot,
                                                  uint32_t run, int nwaves, int wpb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * wpb + wave;
    if (w >= nwaves || wave >= wpb) return;
    uint4* o = reinterpret_cast<uint4*>(base + start[w]);
    const uint32_t nch = run / 16, r0 = rot[w] / 16;
    const uint4 v = make_uint4(w, lane, 3, 4);
    // two linear pieces: [r0, nch) then [0, r0)
    for (int piece = 0; piece < 2; piece++) {
        const uint32_t lo = piece == 0 ? r0 : 0, hi = piece == 0 ? nch : r0;
        uint32_t c = lo + lane;
        for (; c + 192 < hi; c += 256) { o[c] = v; o[c + 64] = v; o[c + 128] = v; o[c + 192] = v; }
        for (; c < hi; c += 64) o[c] = v;
    }
}
__global__ void k_fill(uint4* out, size_t nchunks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) out[i] = make_uint4(1, 2, 3, 4);
}

static hipEvent_t ea, eb;
template <class F> static float timeit(F launch, int iters = 12) {
    launch(); launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(ea));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / iters;
}

struct Pattern { std::vector<uint64_t> start; std::vector<uint32_t> rot; uint32_t run; int wpb; };

int main(int argc, char** argv) {
    const size_t GiB = 1ull << 30, MiB = 1ull << 20;
    const size_t arena_bytes = (argc > 1 ? atoi(argv[1]) : 40) * GiB;
    uint8_t* arena; CK(hipMalloc(&arena, arena_bytes));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    uint64_t* d_start; uint32_t* d_rot;
    CK(hipMalloc(&d_start, 65536 * 8)); CK(hipMalloc(&d_rot, 65536 * 4));
    const uint32_t S = 225792, W = 4096;           // the raster at 32 768 envs: 8 envs x 28 224 B per wave, 4 096 waves
    const double total = (double)S * W;
    printf("arena %p, %zu GiB; S = %u B between streams, %u streams, %.1f MB per launch\n", (void*)arena, arena_bytes / GiB, S, W, total / 1e6);

    auto run = [&](const char* name, uint8_t* base, const Pattern& p) {
        const int nw = (int)p.start.size();
        CK(hipMemcpy(d_start, p.start.data(), nw * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rot, p.rot.data(), nw * 4, hipMemcpyHostToDevice));
        const int blocks = (nw + p.wpb - 1) / p.wpb;
        const float ms = timeit([&] { hipLaunchKernelGGL(k_streams, dim3(blocks), dim3(p.wpb * 64), 0, 0, base, d_start, d_rot, p.run, nw, p.wpb); });
        const double bytes = (double)p.run * nw;
        printf("%-64s %7.4f ms  %6.0f GB/s\n", name, ms, bytes / ms / 1e6);
        fflush(stdout);
        return ms;
    };
    auto linear = [&](uint32_t stride, uint32_t nw = W, uint32_t runb = S, int wpb = 16) {
        Pattern p; p.run = runb; p.wpb = wpb;
        for (uint32_t w = 0; w < nw; w++) { p.start.push_back((uint64_t)w * stride); p.rot.push_back(0); }
        return p;
    };
    uint8_t* A = arena + 1 * GiB;                  // physically contiguous from here on (inside the first 32 GiB block)
    char name[256];

    if (argc > 2 && !strcmp(argv[2], "scan")) {
        // `stream_conflicts <GiB> scan [step MiB]`: the raster's stream set at every step of a big arena — where are the
        // valleys (visit 3: one at the 32 GiB | 8 GiB block boundary of a 40 GiB arena, at fill speed), how far apart, and
        // does a GAP of 16 / 32 / 64 GiB between the two halves of the set — inside ONE physical block — do what the
        // block boundary does?
        const size_t step = (argc > 3 ? atoi(argv[3]) : 512) * MiB;
        const size_t span = (size_t)total + MiB;
        printf("--- scan: baseline pattern every %zu MiB (offset_GiB:GB/s; < 6000 GB/s printed only every 4th)\n", step / MiB);
        int i = 0;
        for (size_t off = 0; off + span < arena_bytes; off += step, i++) {
            Pattern p = linear(S);
            const int nw = (int)p.start.size();
            CK(hipMemcpy(d_start, p.start.data(), nw * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_rot, p.rot.data(), nw * 4, hipMemcpyHostToDevice));
            uint8_t* base = arena + off;
            const float ms = timeit([&] { hipLaunchKernelGGL(k_streams, dim3(256), dim3(1024), 0, 0, base, d_start, d_rot, p.run, nw, 16); }, 4);
            const double gbs = total / ms / 1e6;
            if (gbs >= 5700 || i % 4 == 0) printf("%.2f:%.0f%s ", off / (double)GiB, gbs, gbs >= 5700 ? "*" : "");
        }
        printf("\n--- gaps between the two halves of the set (streams w >= 2048 moved up by D), base at +1 GiB and at +33 GiB\n");
        for (size_t b0 : {1 * GiB, 33 * GiB}) for (double dg : {0.0, 4.0, 8.0, 12.0, 16.0, 24.0, 31.0, 32.0, 33.0, 48.0, 64.0}) {
            if (b0 + (size_t)(dg * GiB) + span >= arena_bytes) continue;
            Pattern p = linear(S);
            for (uint32_t w = 2048; w < W; w++) p.start[w] += (uint64_t)(dg * GiB);
            snprintf(name, sizeof name, "base +%zu GiB, upper half + %.0f GiB", b0 / GiB, dg);
            run(name, arena + b0, p);
        }
        return 0;
    }

    {   // for scale
        const size_t nch = (size_t)total / 16;
        const float f = timeit([&] { hipLaunchKernelGGL(k_fill, dim3((nch + 255) / 256), dim3(256), 0, 0, (uint4*)A, nch); });
        printf("%-64s %7.4f ms  %6.0f GB/s\n", "fill of the same bytes", f, total / f / 1e6);
    }
    printf("--- 0. the raster's stream set at a few places of the contiguous part\n");
    for (size_t off : {0 * GiB, 3 * GiB, 11 * GiB, 20 * GiB}) { snprintf(name, sizeof name, "baseline at +%zu GiB", off / GiB + 1); run(name, A + off, linear(S)); }

    if (arena_bytes >= 40 * GiB) {
        printf("--- 1. across the real block boundary at 32 GiB (is this kernel a proxy of the raster?)\n");
        for (long d : {-882L, -661L, -441L, -220L, 0L}) {
            snprintf(name, sizeof name, "baseline starting %ld MiB before the 32 GiB boundary", -d);
            run(name, arena + 32 * GiB + d * (long)MiB, linear(S));
        }
    }
    printf("--- 2. a GAP in the middle of the stream set: streams w >= 2048 moved up by D (contiguous memory)\n");
    for (double dm : {0.0, 0.25, 0.5, 1.0, 2.0, 3.0, 4.0, 8.0, 16.0, 32.0, 64.0, 128.0, 256.0, 441.0, 512.0, 1024.0, 2048.0, 4096.0, 8192.0, 1000.0, 1234.5}) {
        Pattern p = linear(S);
        const uint64_t D = (uint64_t)(dm * MiB) / 16 * 16;
        for (uint32_t w = 2048; w < W; w++) p.start[w] += D;
        snprintf(name, sizeof name, "upper half + %.2f MiB", dm);
        run(name, A, p);
    }
    printf("--- 2b. gaps at every quarter / every eighth (streams in 4 / 8 groups, each group D further)\n");
    for (double dm : {1.0, 8.0, 64.0, 441.0, 1000.0}) for (int groups : {4, 8, 16}) {
        Pattern p = linear(S);
        const uint64_t D = (uint64_t)(dm * MiB) / 16 * 16;
        for (uint32_t w = 0; w < W; w++) p.start[w] += (uint64_t)(w / (W / groups)) * D;
        snprintf(name, sizeof name, "%d groups, each + %.0f MiB more", groups, dm);
        run(name, A, p);
    }
    printf("--- 3. the stride between streams (run stays %u B; gaps between the runs)\n", S);
    for (uint32_t st : {225792u, 225792u + 256, 225792u + 512, 225792u + 1024, 225792u + 4096, 229376u, 233472u, 245760u, 262144u, 262144u + 256,
                        262144u + 1024, 262144u + 4096, 262144u + 16384, 294912u, 327680u, 524288u, 524288u + 4096}) {
        snprintf(name, sizeof name, "stride %u B (= %.3f x 64 KiB)", st, st / 65536.0);
        run(name, A, linear(st));
    }
    printf("--- 4. rotations: every wave starts r(w) into its own run and wraps (same bytes, same layout)\n");
    {
        auto with_rot = [&](const char* nm, auto f) {
            Pattern p = linear(S);
            for (uint32_t w = 0; w < W; w++) p.rot[w] = (uint32_t)(f(w) % S) / 16 * 16;
            run(nm, A, p);
        };
        with_rot("rot = (w & 1) * S/2", [&](uint32_t w) { return (uint64_t)(w & 1) * (S / 2); });
        with_rot("rot = (w >> 11) * S/2   (upper half of the set half a run ahead)", [&](uint32_t w) { return (uint64_t)(w >> 11) * (S / 2); });
        with_rot("rot = (w & 7) * 28224   (env k first, k = w mod 8)", [&](uint32_t w) { return (uint64_t)(w & 7) * 28224; });
        with_rot("rot = ((w >> 9) & 7) * 28224", [&](uint32_t w) { return (uint64_t)((w >> 9) & 7) * 28224; });
        with_rot("rot = (popcount(w) & 7) * 28224", [&](uint32_t w) { return (uint64_t)(__builtin_popcount(w) & 7) * 28224; });
        with_rot("rot = ((w ^ w>>3 ^ w>>6 ^ w>>9) & 7) * 28224", [&](uint32_t w) { return (uint64_t)((w ^ (w >> 3) ^ (w >> 6) ^ (w >> 9)) & 7) * 28224; });
        with_rot("rot = (w * 2654435761 >> 8) mod S   (pseudo-random)", [&](uint32_t w) { return (uint64_t)((w * 2654435761u) >> 8); });
        with_rot("rot = w * 1024 mod S", [&](uint32_t w) { return (uint64_t)w * 1024; });
        with_rot("rot = w * 4096 mod S", [&](uint32_t w) { return (uint64_t)w * 4096; });
        with_rot("rot = w * S / 4096   (a sweep of phases)", [&](uint32_t w) { return (uint64_t)w * S / 4096; });
        with_rot("rot = bitrev12(w) * S / 4096", [&](uint32_t w) { uint32_t r = 0; for (int i = 0; i < 12; i++) r |= ((w >> i) & 1u) << (11 - i); return (uint64_t)r * S / 4096; });
    }
    printf("--- 5. fewer / more streams over the same bytes\n");
    run("2048 streams x 2 S (8 waves per workgroup)", A, linear(2 * S, 2048, 2 * S, 8));
    run("1024 streams x 4 S (4 waves per workgroup)", A, linear(4 * S, 1024, 4 * S, 4));
    run("8192 streams x S/2 (two rounds of workgroups)", A, linear(S / 2, 8192, S / 2, 16));
    run("4096 streams, 12 waves per workgroup (342 workgroups)", A, linear(S, W, S, 12));
    printf("--- 6. which stream goes to which wave (the set is the same; who is whose neighbour changes)\n");
    {
        Pattern p = linear(S);
        for (uint32_t w = 0; w < W; w++) p.start[w] = (uint64_t)((w * 2731u) & 4095u) * S;
        run("region = w * 2731 mod 4096", A, p);
        for (uint32_t w = 0; w < W; w++) { uint32_t r = 0; for (int i = 0; i < 12; i++) r |= ((w >> i) & 1u) << (11 - i); p.start[w] = (uint64_t)r * S; }
        run("region = bitrev12(w)", A, p);
        for (uint32_t w = 0; w < W; w++) p.start[w] = (uint64_t)(((w & 15u) << 8) | (w >> 4)) * S;     // a workgroup's 16 waves 256 regions apart
        run("region = (w & 15) * 256 + (w >> 4)", A, p);
    }
    printf("--- 7. baseline again\n");
    run("baseline at +1 GiB", A, linear(S));
    return 0;
}

// mg_device.h — device-side helpers shared by the gfx950 kernels of libmarlgrid_hip.so.
//
// Written for CDNA4 only (wave64, 160 KiB LDS/CU); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "marlgrid_hip.h"

namespace mg {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves per workgroup everywhere

// ---- packed agent record (include/marlgrid_hip.h MG_AG_*) -------------------------------------
__device__ __forceinline__ uint32_t rec_byte(uint64_t r, int i) { return (uint32_t)(r >> (8 * i)) & 0xFFu; }
__device__ __forceinline__ uint64_t rec_set(uint64_t r, int i, uint32_t v) {
    return (r & ~(0xFFull << (8 * i))) | ((uint64_t)(v & 0xFFu) << (8 * i));
}
__device__ __forceinline__ uint32_t rec_xy(uint64_t r) { return (uint32_t)r & 0xFFFFu; }  // x | y<<8

// forward vector per dir: agents.py:183  [(1,0),(0,1),(-1,0),(0,-1)]
__device__ __forceinline__ int dir_dx(int d) { return d == 0 ? 1 : (d == 2 ? -1 : 0); }
__device__ __forceinline__ int dir_dy(int d) { return d == 1 ? 1 : (d == 3 ? -1 : 0); }

// ---- per-env MT19937 in *lazy* form ------------------------------------------------------------
// numpy's RandomState regenerates all 624 words when the block is exhausted; the same sequence
// falls out of regenerating word `pos` right before it is consumed (the in-place block loop reads
// exactly the values this does), which needs no 624-iteration "twist" stall in one lane: a draw
// is 3 loads + 1 store.  State after seeding: pos = 0 == numpy's pos 624.
struct Mt {
    uint32_t* w;  // this env's 624 words (HBM)
    int pos;
    __device__ __forceinline__ uint32_t next() {
        int i = pos;
        int i1 = (i + 1 == MG_MT_N) ? 0 : i + 1;
        int im = i + 397;
        if (im >= MG_MT_N) im -= MG_MT_N;
        uint32_t y = (w[i] & 0x80000000u) | (w[i1] & 0x7fffffffu);
        uint32_t v = w[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        w[i] = v;
        pos = i1;
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        return v;
    }
    // numpy legacy masked rejection (RandomState.randint with array bounds / shuffle's
    // random_interval): smallest 2^k-1 >= max; redraw until (w & mask) <= max; max==0 draws nothing
    __device__ __forceinline__ uint32_t bounded(uint32_t max) {
        if (max == 0) return 0;
        uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
};

// intra-wave LDS hand-off: DS operations of one wave execute in order, so only the compiler has
// to be kept from reordering across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// per-wave LDS scratch of the obs-render kernel (bytes), shared by host launch code and kernel
struct RenderScratch {
    int grid, first, second, rec, vbase, vshow, trow, vis, tmap, total;
};
__host__ __device__ inline RenderScratch render_scratch_layout(int cells_stride, int n, int vs) {
    RenderScratch s;
    int o = 0;
    s.grid = o;  o += round_up(cells_stride, 16);
    s.first = o; o += round_up(cells_stride, 16);
    s.second = o; o += round_up(cells_stride, 16);
    s.rec = o;   o += MG_MAX_AGENTS * 8;
    s.vbase = o; o += round_up(n * vs * vs, 16);
    s.vshow = o; o += round_up(n * vs * vs, 16);
    s.trow = o;  o += round_up(n * vs * 4, 16);
    s.vis = o;   o += round_up(n * vs * 4, 16);
    s.tmap = o;  o += round_up(n * vs * vs * 2, 16);
    s.total = o;
    return s;
}

}  // namespace mg

// mg_raster_front.hip — the observation raster as a DENSE MOVING FRONT over the obs tensor.
//
// Why a second raster.  Measured on MI355X (tools/microbench/store_patterns{2..5}.hip, profiles/r02):
// HBM absorbs 6.6 TB/s when, at any moment, the chip writes ONE contiguous ~1 MiB window in aligned
// 4 KiB blocks from four waves per CU (a plain fill: 6.9), but only 5.3-5.5 TB/s when every wave or
// workgroup streams its own contiguous span — whatever the span's size, alignment or parity, the waves
// per CU, the store width or flavour.  mg_render.hip is of the second kind by construction: a wave owns
// whole envs, because deriving an env's view (crop, occlusion, tile selection) is per-env work.  Here
// that coupling is cut: the view kernel (mg_render.hip in views-only mode, with the env step fused in
// front) leaves each env's tmap — n * VS * VS atlas offsets, 2 bytes each — in HBM, and this kernel
// turns tmaps into pixels in pure address order: in sub-round q, workgroup w writes the 4 KiB block
// q * gridDim + w of the tensor, whichever env(s) it belongs to.
//
// Shape: one workgroup per CU, 5 waves.  Waves 0-3 are STREAMERS: lane t of the workgroup owns chunk t
// (16 bytes) of each of the round's 4 blocks; a chunk is two 8-byte pairs, each a ds_read_b64 from the
// LDS-resident atlas at tmap[cell] + row * TD + k, exactly as in mg_render.hip's chunk raster; the 4
// stores are issued back to back.  Wave 4 is the LOADER: it fetches the few dozen tmap entries each block
// of the NEXT round needs (one contiguous run per block: tmaps are laid out [B][n*VS*VS]) into an LDS
// double buffer — a wave that never stores, so that no load ever queues behind the streamers' stores
// (vmcnt is shared and in order on gfx950).  One s_barrier per round hands the buffer over.  The
// (env, chunk-in-env) coordinates of a lane's chunks advance by a constant per round: no division in
// the loop.
#include "mg_device.h"
#include "mg_launch.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif

namespace mg {

constexpr int kFrontSub = 16;      // 4 KiB blocks per workgroup and round: long enough (~2.5 us of streaming) that the
                                   // loader's one-round-ahead fetch is never what the barrier waits for
constexpr int kFrontGroup = 8;     // blocks a streamer lane gathers before it issues their stores back to back
constexpr int kFrontSlice = 64;    // tmap entries staged per block (host checks the geometry fits)

// LDS hand-over: DS operations drained, then the workgroup barrier.  (Not __syncthreads(): its release
// fence would also drain the streamers' global stores — vmcnt(0) — once per round.)
__device__ __forceinline__ void front_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int VS_, int TS_>
__global__ __launch_bounds__(320) void raster_front_kernel(MgConfig cfg, const uint16_t* __restrict__ tmap,
                                                           uint8_t* __restrict__ obs, int ab_mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t TD = TS_ * 3 / 4, PT = TD / 2, PR = VS_ * PT;     // dwords / pairs per tile row, pairs per pixel row
    constexpr uint32_t P = VS_ * TS_, VV = VS_ * VS_;
    const uint32_t n = (uint32_t)cfg.n_agents, NV = n * VV;
    const uint32_t CPE = n * P * PR / 2;                                  // 16-byte chunks per env
    const uint32_t nchunks = (uint32_t)cfg.B * CPE;                       // (host: < 2^31)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t G = gridDim.x, w = blockIdx.x;

    const int atlas_bytes = round_up(4 * cfg.n_tiles * TS_ * TS_ * 3, 16);
    const uint32_t* atlas32 = reinterpret_cast<const uint32_t*>(smem);
    uint16_t* s_slice = reinterpret_cast<uint16_t*>(smem + atlas_bytes);                       // [2][kFrontSub][kFrontSlice]
    uint32_t* s_base = reinterpret_cast<uint32_t*>(s_slice + 2 * kFrontSub * kFrontSlice);     // [2][kFrontSub]
    // geometry table, one entry per chunk of an env: for each of its two pairs the tmap cell (band * VS +
    // column, counted over the env's n images) and the dword offset inside the tile (row * TD + 2 * k)
    uint2* s_geo = reinterpret_cast<uint2*>(s_base + 2 * kFrontSub);                           // [CPE] {cell0 | off0 << 16, cell1 | off1 << 16}
    {
        const uint4* src = reinterpret_cast<const uint4*>(cfg.atlas);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < atlas_bytes / 16; i += 320) dst[i] = src[i];
        for (uint32_t c = tid; c < CPE; c += 320) {
            const uint32_t p0 = 2u * c, r0 = p0 / PR, pr0 = p0 - r0 * PR;
            uint32_t r1 = r0, pr1 = pr0 + 1u;
            if (pr1 == PR) { pr1 = 0; r1++; }
            const uint32_t va0 = pr0 / PT, kp0 = pr0 - va0 * PT, vb0 = r0 / (uint32_t)TS_, rr0 = r0 - vb0 * (uint32_t)TS_;
            const uint32_t va1 = pr1 / PT, kp1 = pr1 - va1 * PT, vb1 = r1 / (uint32_t)TS_, rr1 = r1 - vb1 * (uint32_t)TS_;
            s_geo[c] = make_uint2((vb0 * VS_ + va0) | ((rr0 * TD + kp0 * 2u) << 16), (vb1 * VS_ + va1) | ((rr1 * TD + kp1 * 2u) << 16));
        }
    }
    const uint32_t nblocks = (nchunks + 255u) >> 8;
    const uint32_t per_round = G * kFrontSub;
    const uint32_t rounds = (nblocks + per_round - 1) / per_round;
    const uint32_t stride = per_round << 8;                      // chunks between a lane's (a block's) chunks of consecutive rounds
    const uint32_t qs = stride / CPE, rs = stride - qs * CPE;

    // the loader's work for one round: per block, the contiguous run of tmap entries it needs.  All of the
    // round's loads are issued before the first LDS write (one memory latency per round, not 16), and the
    // blocks' (env, chunk-in-env) coordinates advance by the same constant per round as the streamers':
    // no division by the run-time env size in the loop either.
    uint32_t le[kFrontSub], lci[kFrontSub];                       // first chunk of block j of the loader's next round
    auto load_round = [&](int buf) {
        uint16_t val[kFrontSub];
        uint32_t s0a[kFrontSub];
#pragma unroll
        for (int j = 0; j < kFrontSub; j++) {
            const uint32_t e0 = le[j], ci0 = lci[j];
            const bool inside = e0 < (uint32_t)cfg.B;             // (C0 < nchunks)
            uint32_t e1 = e0, ci1 = ci0 + 255u;
            if (ci1 >= CPE) { ci1 -= CPE; e1++; }
            if (e1 >= (uint32_t)cfg.B) { e1 = (uint32_t)cfg.B - 1u; ci1 = CPE - 1u; }   // the tensor's last, partial block
            const uint32_t vb0 = ((2u * ci0) / PR) / (uint32_t)TS_, vb1 = ((2u * ci1 + 1u) / PR) / (uint32_t)TS_;
            const uint32_t s0 = inside ? e0 * NV + vb0 * VS_ : 0u, s1 = inside ? e1 * NV + vb1 * VS_ + VS_ : 1u;
            s0a[j] = s0;
            val[j] = tmap[(size_t)s0 + min((uint32_t)lane, s1 - s0 - 1u)];
            le[j] += qs;
            lci[j] += rs;
            if (lci[j] >= CPE) { lci[j] -= CPE; le[j]++; }
        }
#pragma unroll
        for (int j = 0; j < kFrontSub; j++) {
            s_slice[(buf * kFrontSub + j) * kFrontSlice + lane] = val[j];
            if (lane == 0) s_base[buf * kFrontSub + j] = s0a[j];
        }
    };

    // a streamer lane's chunk of sub-block j: (env * NV, chunk in env), advanced by a constant per round
    uint32_t eNV[kFrontSub], ci[kFrontSub];
    if (wave == 4) {
#pragma unroll
        for (int j = 0; j < kFrontSub; j++) {
            const uint32_t C = ((uint32_t)j * G + w) << 8;
            le[j] = C / CPE;
            lci[j] = C - le[j] * CPE;
        }
    }
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < kFrontSub; j++) {
            const uint32_t C = (((uint32_t)j * G + w) << 8) + (uint32_t)tid;
            const uint32_t e = C / CPE;
            eNV[j] = e * NV;
            ci[j] = C - e * CPE;
        }
    }
    if (wave == 4) load_round(0);
    __syncthreads();                                             // atlas + round 0 slices

    for (uint32_t R = 0; R < rounds; R++) {
        const int buf = (int)(R & 1u);
        if (wave == 4) {
#if defined(MG_AB_VARIANTS)
            if (ab_mode != 6)
#endif
            if (R + 1 < rounds) load_round(buf ^ 1);
        }
#if defined(MG_AB_VARIANTS)
        else if (ab_mode >= 3 && ab_mode <= 6) {     // (measurement only: the store pattern alone, constant data;
            uint4* out = reinterpret_cast<uint4*>(obs);   //  4: every workgroup writes 16 ADJACENT blocks per round)
#pragma unroll
            for (int j = 0; j < kFrontSub; j++) {
                const size_t blk = ab_mode == 3 ? (size_t)(R * kFrontSub + (uint32_t)j) * G + w
                                                : (size_t)R * kFrontSub * G + (size_t)w * kFrontSub + (size_t)j;
                const size_t C = (blk << 8) + (size_t)tid;
                if (C < nchunks) out[C] = make_uint4(R, j, 3, 4);
            }
        }
#endif
        else {
            // branch-free: chunks past the end (last round only) are computed on clamped coordinates and
            // just not stored, so the chunks' LDS chains (geometry -> tmap -> atlas) overlap
            uint4* out = reinterpret_cast<uint4*>(obs);
#pragma unroll
            for (int j0 = 0; j0 < kFrontSub; j0 += kFrontGroup) {
                uint4 v[kFrontGroup];
                bool on[kFrontGroup];
                uint2 geo[kFrontGroup];
                uint32_t rel[kFrontGroup], t0[kFrontGroup], t1[kFrontGroup];
#pragma unroll
                for (int g = 0; g < kFrontGroup; g++) {
                    const int j = j0 + g;
                    const uint32_t C = (((R * kFrontSub + (uint32_t)j) * G + w) << 8) + (uint32_t)tid;
                    on[g] = C < nchunks;
                    geo[g] = s_geo[ci[j]];
                    rel[g] = on[g] ? eNV[j] - s_base[buf * kFrontSub + j] : 0u;
                }
#pragma unroll
                for (int g = 0; g < kFrontGroup; g++) {
                    const uint16_t* sl = s_slice + (buf * kFrontSub + j0 + g) * kFrontSlice;
                    t0[g] = sl[(rel[g] + (geo[g].x & 0xFFFFu)) & (kFrontSlice - 1)];
                    t1[g] = sl[(rel[g] + (geo[g].y & 0xFFFFu)) & (kFrontSlice - 1)];
                }
#pragma unroll
                for (int g = 0; g < kFrontGroup; g++) {
                    const int j = j0 + g;
                    const uint2 a = *reinterpret_cast<const uint2*>(atlas32 + ((t0[g] + (geo[g].x >> 16)) & 0x7FFEu));
                    const uint2 c = *reinterpret_cast<const uint2*>(atlas32 + ((t1[g] + (geo[g].y >> 16)) & 0x7FFEu));
                    v[g] = make_uint4(a.x, a.y, c.x, c.y);
#if defined(MG_AB_VARIANTS)
                    if (ab_mode == 1) v[g] = make_uint4(t0[g], t1[g], geo[g].x, geo[g].y);   // no atlas gathers
#endif
                    eNV[j] += qs * NV;                                        // next round
                    ci[j] += rs;
                    if (ci[j] >= CPE) { ci[j] -= CPE; eNV[j] += NV; }
                }
#pragma unroll
                for (int g = 0; g < kFrontGroup; g++) {
                    const size_t C = ((size_t)((R * kFrontSub + (uint32_t)(j0 + g)) * G + w) << 8) + (size_t)tid;
                    if (on[g]) out[C] = v[g];
                }
            }
        }
#if defined(MG_AB_VARIANTS)
        if (ab_mode == 2 || ab_mode == 5 || ab_mode == 6) continue;     // (measurement only: no hand-over barrier)
#endif
        front_barrier();
    }
}

// Can this configuration take the two-kernel form?  The chunk raster's tile sizes with the atlas resident
// in LDS, static agent colours, an env of at least one block (a block then touches at most two envs), a
// per-block tmap run that fits the staging slots, 32-bit chunk indices, and enough blocks to fill the chip.
bool raster_front_eligible(const MgConfig& cfg) {
    const int vs = cfg.view_size, ts = cfg.tile_size, n = cfg.n_agents;
    if (!((ts == 8 && (vs == 7 || vs == 9 || vs == 5)) || (ts == 16 && vs == 7))) return false;
    if (cfg.prestige_mask) return false;
    const long long atlas_b = round_up(4 * cfg.n_tiles * ts * ts * 3, 16);
    const long long P = (long long)vs * ts, S = (long long)n * P * P * 3, cpe = S / 16;
    if (atlas_b + 4096 + cpe * 8 > 128 * 1024 || atlas_b / 4 > 0x7FFE) return false;   // atlas + geometry table in LDS
    if (cpe < 256 || (long long)cfg.B * cpe >= (1ll << 31)) return false;
    const long long band_bytes = P * 3 * ts;
    if ((4096 / band_bytes + 2) * vs > kFrontSlice) return false;
    return (long long)cfg.B * cpe >= 256ll * 256 * kFrontSub * 2; // at least two full rounds of the whole chip
}

template <int VS_, int TS_>
static hipError_t launch_front_t(const MgConfig& cfg, const uint16_t* tmap, uint8_t* obs, hipStream_t s) {
    const long long P = (long long)VS_ * TS_, cpe = (long long)cfg.n_agents * P * P * 3 / 16;
    const size_t lds = (size_t)round_up(4 * cfg.n_tiles * TS_ * TS_ * 3, 16) + 2 * kFrontSub * kFrontSlice * 2 + 2 * kFrontSub * 4 +
                       (size_t)cpe * 8;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raster_front_kernel<VS_, TS_>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int ab_mode = 0;
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_FRONT_MODE")) ab_mode = atoi(f);
#endif
    hipLaunchKernelGGL((raster_front_kernel<VS_, TS_>), dim3(256), dim3(320), lds, s, cfg, tmap, obs, ab_mode);
    return hipGetLastError();
}

hipError_t launch_raster_front(const MgConfig& cfg, const uint16_t* tmap, uint8_t* obs, hipStream_t s) {
    const int vs = cfg.view_size, ts = cfg.tile_size;
    if (ts == 8 && vs == 7) return launch_front_t<7, 8>(cfg, tmap, obs, s);
    if (ts == 8 && vs == 9) return launch_front_t<9, 8>(cfg, tmap, obs, s);
    if (ts == 8 && vs == 5) return launch_front_t<5, 8>(cfg, tmap, obs, s);
    if (ts == 16 && vs == 7) return launch_front_t<7, 16>(cfg, tmap, obs, s);
    return hipErrorInvalidValue;
}

}  // namespace mg

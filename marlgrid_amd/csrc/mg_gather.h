// mg_gather.h — the GATHER raster of the observation kernel (mg_render_kernel.h, RM_ == 2): tile sizes whose rows
// are not whole 8-byte pairs — the reference's default view_tile_size 5 (agents.py:22), 6, 7, 9, 10, 11, 12 at the
// default view size 7, and views 3 .. 9 at 5-pixel tiles — rendered OUTPUT-centric: a lane composes one aligned 16-byte chunk of the output stream in
// registers and stores it.
//
// What is rendered (MultiGrid.render, base.py:301-331): an env's n images are P = VS * TS pixel rows of RB = 3 * P bytes
// each, back to back — i.e. ONE stream of SEGMENTS of SEG = 3 * TS bytes (one pixel row of one view cell's tile),
// VS per pixel row, and the envs of a group follow each other without a gap, so a whole group of envs is one
// stream of segments whose tile rows are looked up through the group's tmap ([band][column], dense: band g of the
// group is pixel rows [g * TS, (g + 1) * TS)).  SEG >= 15, so an aligned 16-byte chunk of that stream touches at
// most TWO consecutive segments, A and B:
//     chunk = window(A, k0) | window(B, k0 - SEG)        window(S, k) = bytes [k, k + 16) of S's tile row, 0 outside it
// The tile rows sit in LDS padded with zeros — 16 zero bytes, the SEG bytes, zeros up to RS (a multiple of 4) —
// so a window is five ALIGNED dwords read around the row (an unaligned DS access is serialised on gfx950) cut to
// the chunk's phase with four v_alignbyte: nothing is masked, nothing is conditional, no byte crosses LDS twice
// (the assemble-and-stream raster ORs every segment into a zeroed piece buffer with LDS atomics and then streams
// the buffer out: ~45 instructions per 15 bytes plus the second pass).
//
// The lane -> chunk mapping is periodic: PC = RB / gcd(16, RB) chunks are PR = 16 / gcd(16, RB) whole pixel
// rows, so a lane that always takes the same place in the period has its segment columns, its offset k0 in
// segment A and both windows' alignments as CONSTANTS, worked out once per group of envs (gather_lane); a trip only
// adds the row.  Tile 6: 63 chunks = 8 rows — 63 lanes, and three periods are four whole bands of tiles, so with three
// sets of constants even the tile row inside the band is constant: a window is add, tmap look-up, multiply-add, five
// dwords.  Tile 5: 105 chunks = 16 rows in two trips of 53 + 52 lanes; the band and the tile row are worked out
// per trip (five periods of constants do not fit the registers).  Tile 11: 231 chunks in four trips of 58 lanes, two
// and two.  'prestige' agents (the reference's example, tile 11): their per-env recoloured tiles are virtual tiles
// behind the atlas's, in the same padded layout in the wave's own scratch (GatherDyn).
//
// Plain inline functions over byte pointers: the same text compiles with g++ for tests/native (the index
// arithmetic is checked on the host against a byte-by-byte raster; the product never loads that build).
#pragma once

#include <stdint.h>

#include "mg_core.h"

// (host harness: tests/native defines this to check every LDS offset the raster forms against the buffers' sizes)
#if !defined(MG_GATHER_BOUNDS)
#define MG_GATHER_BOUNDS(tmap_off, atlas_off) do {} while (0)
#endif

namespace mg {

MG_HD uint32_t gather_align(uint32_t hi, uint32_t lo, uint32_t sh) {   // bytes [sh, sh + 4) of lo | hi << 32, sh = 0..3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> (8u * (sh & 3u)));
#endif
}

constexpr int gather_gcd(int a, int b) { return b ? gather_gcd(b, a % b) : a; }

// geometry of one (view size, tile size)
template <int VS, int TS>
struct GatherGeom {
    static_assert(TS >= 5, "a 16-byte chunk must touch at most two segments: 3 * TS >= 15");
    static constexpr int SEG = 3 * TS, RB = VS * SEG;
    static constexpr int FRONT = 16;                              // zero bytes in front of a padded tile row
    static constexpr int RS = (FRONT + SEG + 3) / 4 * 4;          // bytes per padded tile row (the zeros behind it: the next row's front)
    static constexpr int TILE = TS * RS;                          // bytes per padded tile
    static constexpr int TAIL = 32;                               // zero bytes behind the last row of the atlas
    static constexpr int G16 = gather_gcd(16, RB);
    static constexpr int PC = RB / G16, PR = 16 / G16;            // chunks / pixel rows per period
    // a CYCLE of C periods = NT <= 4 trips (= sets of lane constants): the C whose trips are best filled, whole bands of
    // tiles (C * PR a multiple of TS: the tile row of a lane's segment is then a constant of the set — a quarter fewer
    // instructions per window) counting for 1.3; ties: the smaller C
    static constexpr int pick_c() {
        int best = 1, best_score = 0;
        for (int c = 1; c <= 8; c++) {
            const int cc = c * PC, nt = (cc + 63) / 64;
            if (nt > 4) continue;
            const int score = cc * 1000 / (nt * 64) * (((c * PR) % TS == 0) ? 13 : 10);
            if (score > best_score) { best = c; best_score = score; }
        }
        return best;
    }
    static constexpr int C = pick_c();
    static constexpr int CC = C * PC;                             // chunks per cycle
    static constexpr int NT = (CC + 63) / 64;                     // trips (= sets of lane constants) per cycle
    static_assert(NT <= 4, "too many sets of lane constants");
    static constexpr int LPT = (CC + NT - 1) / NT;                // lanes of a trip (the last trip of a cycle may have fewer)
    static constexpr int CYC_ROWS = C * PR;                       // pixel rows per cycle
    static constexpr bool kConstBand = (CYC_ROWS % TS) == 0;
    static constexpr int CYC_BANDS = CYC_ROWS / TS;               // (kConstBand)
    static constexpr uint32_t M_TS = ((1u << 20) + TS - 1) / TS;  // row / TS as (row * M_TS) >> 20: exact for row < 2^20 / TS
    static constexpr int lanes_of(int t) { return CC - t * LPT < LPT ? CC - t * LPT : LPT; }
};

// Padded atlas in LDS: dword d of it as a window of the atlas in HBM ([virtual tile][TS][SEG] bytes, `raw16` bytes
// rounded up to 16).  Returns the byte offset of 8 aligned source bytes (-1: the dword is zeros), `cut` = the bit the
// dword starts at in them, `keep` = the mask of its bytes that belong to the row.  FRONT_W: zero dwords in front of a row
// (4 here; the assemble-and-stream raster's padded rows have 1), ROW_W: dwords per padded row.
template <int SEG, int FRONT_W, int ROW_W>
MG_HD int pad_source(int d, int rows, int raw16, uint32_t& cut, uint32_t& keep) {
    const int row = d / ROW_W, k = d - row * ROW_W, j0 = 4 * (k - FRONT_W);    // row bytes [j0, j0 + 4)
    const int nvb = SEG - j0 < 4 ? SEG - j0 : 4;
    cut = 0;
    keep = 0;
    if (k < FRONT_W || nvb <= 0 || row >= rows) return -1;
    const int sb = row * SEG + j0;
    int a = sb & ~3;
    if (a > raw16 - 8) a = raw16 - 8;
    cut = (uint32_t)(sb - a) * 8u;                                              // (0 .. 56 bits)
    keep = nvb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nvb)) - 1u;
    return a;
}
MG_HD uint32_t pad_cut(uint32_t lo, uint32_t hi, uint32_t cut, uint32_t keep) {
    return (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> cut) & keep;
}

// A lane's constants, per set (= trip of the cycle).  Window A: the segment the chunk starts in; B: the next one.
//   t*: byte offset of the segment's column in the tmap — kConstBand: of its (band, column) entry within the cycle
//   s*: byte offset of the window's first aligned dword within the padded tile — kConstBand: tile row included
//   h*: v_alignbyte shift;  r*: pixel row within the cycle (only when the band is worked out per trip)
template <int NT>
struct GatherLane {
    uint32_t ta[NT], sa[NT], ha[NT], ra[NT];
    uint32_t tb[NT], sb[NT], hb[NT], rb[NT];
};

// `qs`: stream byte the group's first whole chunk starts at (0..15).
template <int VS, int TS>
MG_HD GatherLane<GatherGeom<VS, TS>::NT> gather_lane(int lane, uint32_t qs) {
    typedef GatherGeom<VS, TS> Gm;
    GatherLane<Gm::NT> c;
#pragma unroll
    for (int t = 0; t < Gm::NT; t++) {
        const uint32_t slot = (uint32_t)(t * Gm::LPT + lane);
        const uint32_t u = qs + 16u * (slot < (uint32_t)Gm::CC ? slot : 0u);           // (idle lanes: any valid place)
        const uint32_t rowA = u / (uint32_t)Gm::RB, x = u - rowA * Gm::RB;
        const uint32_t colA = x / (uint32_t)Gm::SEG, k0 = x - colA * Gm::SEG;
        const uint32_t offA = Gm::FRONT + k0, d = Gm::SEG - k0;
        // d >= 16: the chunk ends in A.  B then stands for A's own tile row read at its 16 zero bytes in front (the
        // segment behind A may lie behind the group's last: no tmap entry to look at)
        const uint32_t offB = d < 16u ? Gm::FRONT - d : 0u;
        uint32_t colB = colA, rowB = rowA;
        if (d < 16u && ++colB == (uint32_t)VS) { colB = 0; rowB = rowA + 1; }
        c.ha[t] = offA & 3u;
        c.hb[t] = offB & 3u;
        if (Gm::kConstBand) {
            const uint32_t bA = rowA / TS, bB = rowB / TS;
            c.ta[t] = (bA * VS + colA) * 2u;
            c.tb[t] = (bB * VS + colB) * 2u;
            c.sa[t] = (rowA - bA * TS) * Gm::RS + (offA & ~3u);
            c.sb[t] = (rowB - bB * TS) * Gm::RS + (offB & ~3u);
            c.ra[t] = c.rb[t] = 0;
        } else {
            c.ta[t] = colA * 2u;
            c.tb[t] = colB * 2u;
            c.sa[t] = offA & ~3u;
            c.sb[t] = offB & ~3u;
            c.ra[t] = rowA;
            c.rb[t] = rowB;
        }
    }
    return c;
}

// Source address (byte offset from the padded atlas) of one window in cycle `y`.
//   tmap: the group's dense tmap (virtual tile index per [band][column]);  kConstBand: tcyc = y * CYC_BANDS * VS * 2,
//   else rowbase = y * CYC_ROWS
template <int VS, int TS>
MG_HD uint32_t gather_tmap_off(uint32_t t_, uint32_t r_, uint32_t tcyc, uint32_t rowbase, uint32_t& rr) {
    typedef GatherGeom<VS, TS> Gm;
    if (Gm::kConstBand) { rr = 0; return tcyc + t_; }
    const uint32_t R = rowbase + r_;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t band = __umul24(R, Gm::M_TS) >> 20;
#else
    const uint32_t band = (uint32_t)(((uint64_t)R * Gm::M_TS) >> 20);
#endif
    rr = R - band * TS;
    return band * (2u * VS) + t_;
}

// One byte of the group's stream, the slow way (the < 16 bytes in front of the first and behind the last whole chunk).
// kDyn: virtual tiles >= dyn.first are not part of the atlas but a wave's own (the per-env recoloured 'prestige'
// tiles, in the same padded layout): tile t of them lies at atlas + t * TILE + dyn.delta
struct GatherDyn { uint32_t first, delta; };
template <int VS, int TS, bool kDyn = false>
MG_HD uint8_t gather_byte(const uint8_t* tmap, const uint8_t* atlas, uint32_t q, GatherDyn dyn = GatherDyn{0, 0}) {
    typedef GatherGeom<VS, TS> Gm;
    const uint32_t g = q / Gm::SEG, k = q - g * Gm::SEG, R = g / VS, col = g - R * VS, band = R / TS, rr = R - band * TS;
    const uint32_t vt = *reinterpret_cast<const uint16_t*>(tmap + (band * VS + col) * 2u);
    const uint32_t a = vt * Gm::TILE + rr * Gm::RS + Gm::FRONT + k + ((kDyn && vt >= dyn.first) ? dyn.delta : 0u);
    MG_GATHER_BOUNDS((band * VS + col) * 2u, a - 16u);
    return atlas[a];
}

#if defined(__HIP_DEVICE_COMPILE__)
#define MG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MG_SCHED_FENCE() do {} while (0)
#endif

struct alignas(16) GatherChunk { uint32_t x, y, z, w; };     // one aligned 16-byte store

// One cycle (NT trips) of a lane: chunk y * CC + t * LPT + lane for every set t, in three phases with nothing scheduled
// across them — all tmap look-ups, all window reads, then compose and store — for up to three sets at a time (four
// sets: two and two; their forty window registers do not fit a 16-wave workgroup's 128).  kTail: the group's last,
// incomplete cycle — chunks >= nfull are not touched (their tmap entries lie behind the group's).
template <int VS, int TS, bool kTail, bool kDyn = false>
MG_HD void gather_cycle(const GatherLane<GatherGeom<VS, TS>::NT>& c, int lane, uint32_t y, const uint8_t* tmap,
                        const uint8_t* atlas, GatherChunk* out, uint32_t nfull, GatherDyn dyn = GatherDyn{0, 0}) {
    typedef GatherGeom<VS, TS> Gm;
    constexpr int NT = Gm::NT, SB = NT <= 3 ? NT : 2;           // sets per batch
    const uint32_t tcyc = y * (uint32_t)(Gm::CYC_BANDS * VS * 2), rowbase = y * (uint32_t)Gm::CYC_ROWS;
    const uint32_t i0 = y * (uint32_t)Gm::CC + (uint32_t)lane;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += SB) {
        bool on[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) on[u] = lane < Gm::lanes_of(t0 + u) && (!kTail || i0 + (uint32_t)((t0 + u) * Gm::LPT) < nfull);
        uint32_t pa[SB], pb[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int t = t0 + u;
            pa[u] = pb[u] = 0;
            if (!kTail || on[u]) {
                uint32_t rra, rrb;
                const uint32_t oa = gather_tmap_off<VS, TS>(c.ta[t], c.ra[t], tcyc, rowbase, rra);
                const uint32_t ob = gather_tmap_off<VS, TS>(c.tb[t], c.rb[t], tcyc, rowbase, rrb);
                pa[u] = rra * Gm::RS + c.sa[t];
                pb[u] = rrb * Gm::RS + c.sb[t];
                const uint32_t va = *reinterpret_cast<const uint16_t*>(tmap + oa), vb = *reinterpret_cast<const uint16_t*>(tmap + ob);
                pa[u] += va * (uint32_t)Gm::TILE + ((kDyn && va >= dyn.first) ? dyn.delta : 0u);
                pb[u] += vb * (uint32_t)Gm::TILE + ((kDyn && vb >= dyn.first) ? dyn.delta : 0u);
                MG_GATHER_BOUNDS(oa, pa[u]);
                MG_GATHER_BOUNDS(ob, pb[u]);
            }
        }
        MG_SCHED_FENCE();
        uint32_t wa[SB][5], wb[SB][5];
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const uint32_t* qa = reinterpret_cast<const uint32_t*>(atlas + pa[u]);
            const uint32_t* qb = reinterpret_cast<const uint32_t*>(atlas + pb[u]);
#pragma unroll
            for (int j = 0; j < 5; j++) { wa[u][j] = qa[j]; wb[u][j] = qb[j]; }
        }
        MG_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int t = t0 + u;
            if (on[u]) {
                GatherChunk v;
                v.x = gather_align(wa[u][1], wa[u][0], c.ha[t]) | gather_align(wb[u][1], wb[u][0], c.hb[t]);
                v.y = gather_align(wa[u][2], wa[u][1], c.ha[t]) | gather_align(wb[u][2], wb[u][1], c.hb[t]);
                v.z = gather_align(wa[u][3], wa[u][2], c.ha[t]) | gather_align(wb[u][3], wb[u][2], c.hb[t]);
                v.w = gather_align(wa[u][4], wa[u][3], c.ha[t]) | gather_align(wb[u][4], wb[u][3], c.hb[t]);
                out[i0 + (uint32_t)(t * Gm::LPT)] = v;
            }
        }
        MG_SCHED_FENCE();
    }
}

// The raster of a GROUP of envs by one wave: `stream_bytes` bytes at `dst` (any alignment) — whole aligned chunks by
// gather_cycle, the bytes in front of the first and behind the last whole chunk one by one (the chunk they lie in is
// shared with the wave, or the group, before / after: everybody stores its own bytes).
template <int VS, int TS, bool kDyn = false>
MG_HD void gather_group(int lane, const uint8_t* tmap, const uint8_t* atlas, uint8_t* dst, uint32_t stream_bytes,
                        GatherDyn dyn = GatherDyn{0, 0}) {
    typedef GatherGeom<VS, TS> Gm;
    const uint32_t qs = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    const uint32_t nfull = (stream_bytes - qs) >> 4, tail0 = qs + (nfull << 4);
    if ((uint32_t)lane < qs) dst[lane] = gather_byte<VS, TS, kDyn>(tmap, atlas, (uint32_t)lane, dyn);
    if (lane < 16 && tail0 + (uint32_t)lane < stream_bytes) dst[tail0 + lane] = gather_byte<VS, TS, kDyn>(tmap, atlas, tail0 + (uint32_t)lane, dyn);
    const GatherLane<Gm::NT> c = gather_lane<VS, TS>(lane, qs);
    GatherChunk* out = reinterpret_cast<GatherChunk*>(dst + qs);
    uint32_t y = 0;
    for (; (y + 1u) * (uint32_t)Gm::CC <= nfull; y++) gather_cycle<VS, TS, false, kDyn>(c, lane, y, tmap, atlas, out, nfull, dyn);
    if (y * (uint32_t)Gm::CC < nfull) gather_cycle<VS, TS, true, kDyn>(c, lane, y, tmap, atlas, out, nfull, dyn);
}

}  // namespace mg

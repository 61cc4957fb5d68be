// mg_occlude.h — shadow casting (marlgrid/agents.py:298-343) as row bit-masks, shared by the obs
// raster (mg_render.hip) and the whole-grid frame kernel (mg_frame.hip).
#pragma once
#include "mg_device.h"

namespace mg {

// ---- shadow casting (agents.py:298-343) as row bit-masks ---------------------------------------
// bit i of row j == mask[i, j].  The reference sweeps each row rightwards from the agent column
// and leftwards from agent column + 1, propagating to the row above (first loop nest) or below
// (second); out-of-range accesses of the unchecked numba code read False / are dropped.
// W: the row width the flood has to cover (the view size when it is a compile-time constant, else
// MG_MAX_VIEW): log2-many doubling steps — 3 for a 7-wide view, 5 for the 31 columns a 32-bit row mask holds.
template <int W>
__device__ __forceinline__ uint32_t flood_right(uint32_t m, uint32_t p) {
    // set bit i+1 whenever bit i is set and p[i] (p = transparency restricted to [ax, vs-2])
    m |= (m & p) << 1;
    uint32_t q = p & (p >> 1);
    m |= (m & q) << 2;
    if (W > 4) {
        q = q & (q >> 2);
        m |= (m & q) << 4;
    }
    if (W > 8) {
        q = q & (q >> 4);
        m |= (m & q) << 8;
    }
    if (W > 16) {
        q = q & (q >> 8);
        m |= (m & q) << 16;
    }
    return m;
}
template <int W>
__device__ __forceinline__ uint32_t flood_left(uint32_t m, uint32_t p) {
    // set bit i-1 whenever bit i is set and p[i] (p = transparency restricted to [1, ax+1])
    m |= (m & p) >> 1;
    uint32_t q = p & (p << 1);
    m |= (m & q) >> 2;
    if (W > 4) {
        q = q & (q << 2);
        m |= (m & q) >> 4;
    }
    if (W > 8) {
        q = q & (q << 4);
        m |= (m & q) >> 8;
    }
    if (W > 16) {
        q = q & (q << 8);
        m |= (m & q) >> 16;
    }
    return m;
}

// views up to kRegView rows keep the row masks in registers (the loops below are unrolled over them); larger ones — up to
// MG_MAX_VIEW = 31 columns of a 32-bit mask — walk them in memory (occlude_rows_mem)
constexpr int kRegView = 15;
template <int VS_>
__device__ __forceinline__ void occlude_rows(int vs_rt, int off, const uint32_t* __restrict__ T,
                                             uint32_t* __restrict__ out) {
    const int VS = VS_ ? VS_ : vs_rt;
    constexpr int N = VS_ ? VS_ : kRegView;
    const int ax = VS / 2, ay = VS - 1 - off;
    const uint32_t full = (1u << VS) - 1u;
    const uint32_t hi = full & ~((1u << ax) - 1u);            // columns ax .. VS-1
    const uint32_t lo = ((1u << (ax + 2)) - 2u) & full;       // columns 1 .. ax+1
    const uint32_t pr = hi & ~(1u << (VS - 1));               // right flood sources: ax .. VS-2
    uint32_t m[N], t[N];
#pragma unroll
    for (int j = 0; j < N; j++) { m[j] = 0; t[j] = (j < VS) ? T[j] : 0u; }
#pragma unroll
    for (int j = 0; j < N; j++) if (j == ay) m[j] = 1u << ax;
    // first nest: rows ay+1 .. 1 propagate upwards (row ay+1 is still empty there: a no-op)
#pragma unroll
    for (int j = N - 1; j >= 1; j--) {
        if (j < VS && j <= ay) {
            uint32_t r = flood_right<N>(m[j], t[j] & pr);
            uint32_t s = r & t[j] & hi;
            m[j - 1] |= (s | (s << 1)) & full;
            r = flood_left<N>(r, t[j] & lo);
            s = r & t[j] & lo;
            m[j - 1] |= s | (s >> 1);
            m[j] = r;
        }
    }
    // second nest: rows ay .. VS-1 propagate downwards
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (j < VS && j >= ay) {
            uint32_t r = flood_right<N>(m[j], t[j] & pr);
            uint32_t s = r & t[j] & hi;
            uint32_t down = (s | (s << 1)) & full;
            r = flood_left<N>(r, t[j] & lo);
            s = r & t[j] & lo;
            down |= s | (s >> 1);
            m[j] = r;
            if (j + 1 < N && j + 1 < VS) m[j + 1] |= down;
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) if (j < VS) out[j] = m[j];
}

// The same cast with the rows where they are: T[VS] the transparency rows (read only), M[VS] the result (any memory that is
// not T: LDS in the obs kernel, a local array in the frame kernel) — plain loops, no register arrays: view sizes 16 .. 31.
__device__ inline void occlude_rows_mem(int VS, int off, const uint32_t* T, uint32_t* M) {
    const int ax = VS / 2, ay = VS - 1 - off;
    const uint32_t full = (1u << VS) - 1u;
    const uint32_t hi = full & ~((1u << ax) - 1u);            // columns ax .. VS-1
    const uint32_t lo = ((1u << (ax + 2)) - 2u) & full;       // columns 1 .. ax+1
    const uint32_t pr = hi & ~(1u << (VS - 1));               // right flood sources: ax .. VS-2
    for (int j = 0; j < VS; j++) M[j] = j == ay ? 1u << ax : 0u;
    for (int j = ay; j >= 1; j--) {                           // first nest: rows ay .. 1 propagate upwards
        const uint32_t t = T[j];
        uint32_t r = flood_right<MG_MAX_VIEW>(M[j], t & pr);
        uint32_t s = r & t & hi;
        uint32_t up = (s | (s << 1)) & full;
        r = flood_left<MG_MAX_VIEW>(r, t & lo);
        s = r & t & lo;
        up |= s | (s >> 1);
        M[j] = r;
        M[j - 1] |= up;
    }
    for (int j = ay; j < VS; j++) {                           // second nest: rows ay .. VS-1 propagate downwards
        const uint32_t t = T[j];
        uint32_t r = flood_right<MG_MAX_VIEW>(M[j], t & pr);
        uint32_t s = r & t & hi;
        uint32_t down = (s | (s << 1)) & full;
        r = flood_left<MG_MAX_VIEW>(r, t & lo);
        s = r & t & lo;
        down |= s | (s >> 1);
        M[j] = r;
        if (j + 1 < VS) M[j + 1] |= down;
    }
}

}  // namespace mg

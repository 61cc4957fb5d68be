// mg_frame.hip — whole-grid human view of selected envs: the grid half of MultiGridEnv.render
// (marlgrid/base.py:714-759 -> MultiGrid.render with top_agent=None, base.py:301-331), i.e. every
// cell drawn at `frame_tile_size` pixels in world orientation, plus the "visible to some agent"
// highlight (base.py:740-753, 326-329).  Caller-side format: used for
// videos / debugging on a handful of envs, not on the step path.
//
// One workgroup per selected env.  Wave 0 re-derives each active agent's visibility exactly like
// the obs kernel (same crop / rotate / shadow-cast) and scatters it back to world cells; all waves
// then raster rows of the frame as coalesced dwords straight from the (L2-resident) frame atlas.
#include "mg_device.h"
#include "mg_launch.h"
#include "mg_occlude.h"

namespace mg {

__global__ __launch_bounds__(kBlock) void frame_kernel(MgConfig cfg, MgState st, const int32_t* __restrict__ env_ids,
                                                       const uint8_t* __restrict__ atlas, int ts, int highlight,
                                                       uint32_t amax4, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int W = cfg.W, H = cfg.H, n = cfg.n_agents, VS = cfg.view_size;
    const int cells = W * H;
    // LDS: TWO bytes per cell — the cell's tile index, bit 15 = "some agent sees it" — so that every grid a uint8 coordinate
    // addresses fits (255 x 255: 127 KiB; with a copy of the grid, a first-agent map and a highlight plane beside it, 5 bytes per
    // cell, the frame ended at ~180 x 180).  The grid itself is read where it lives (twice per cell: L2).
    uint16_t* s_tile = reinterpret_cast<uint16_t*>(smem);                      // [cells] tile index | kSeen
    uint64_t* s_rec = reinterpret_cast<uint64_t*>(s_tile + round_up(cells, 8));
    uint32_t* s_trow = reinterpret_cast<uint32_t*>(s_rec + MG_MAX_AGENTS);     // [n][VS]
    uint8_t* s_dyn = reinterpret_cast<uint8_t*>(s_trow + round_up(n * VS, 4));   // [n][ts*ts*3] recoloured tiles
    constexpr uint32_t kSeen = 0x8000u;
    const int tid = threadIdx.x;
    const int e = env_ids[blockIdx.x];
    if (e < 0 || e >= cfg.B) return;
    const uint8_t* __restrict__ g_grid = st.grid + (size_t)e * cfg.cells_stride;
    const uint32_t n_tiles = (uint32_t)cfg.n_tiles;

    // tile per world cell: render_tile(obj, top_agent=None) — base.py:275-299.  Every cell its object's plain tile ...
    for (int c = tid; c < cells; c += kBlock) s_tile[c] = (uint16_t)(1u + g_grid[c]);
    if (tid < n) s_rec[tid] = st.agents[(size_t)e * n + tid];
    for (int i = tid; i < n * VS; i += kBlock) s_trow[i] = 0;
    __syncthreads();
    // ... then the lowest-rank agent of a cell puts its own there (the cell object, or `obj.agents[0]` on an overlappable one)
    if (tid < n) {
        const uint64_t r = s_rec[tid];
        if (rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED) {
            bool lowest = true;
            for (int j = 0; j < n; j++) {
                const uint64_t rj = s_rec[j];
                if (j != tid && (rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == rec_xy(r) &&
                    rec_byte(rj, MG_AG_RANK) < rec_byte(r, MG_AG_RANK))
                    lowest = false;
            }
            const int c = (int)rec_byte(r, MG_AG_X) * H + (int)rec_byte(r, MG_AG_Y);
            const uint32_t base = g_grid[c];
            const uint32_t slot = base ? cfg.obj[base].ovl_slot : 0u;
            if (lowest && slot != 0xFF) {
                uint32_t tile;
                if (((cfg.prestige_mask >> tid) & 1u) && (rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE)) tile = n_tiles + (uint32_t)tid;   // dynamic tile (LDS)
                else tile = 1 + cfg.n_obj + (slot * n + (uint32_t)tid) * 4 + rec_byte(r, MG_AG_DIR);
                s_tile[c] = (uint16_t)tile;
            }
        }
    }
    __syncthreads();
    if (highlight) {
        // transparency rows of every agent's view (as in the obs kernel, phase 3)
        const int h = VS / 2, off = cfg.view_offset;
        auto world = [&](const uint64_t r, int va, int vb, int& wx, int& wy) {
            const int x = (int)rec_byte(r, MG_AG_X), y = (int)rec_byte(r, MG_AG_Y), dir = (int)rec_byte(r, MG_AG_DIR);
            if (dir == 3)      { wx = x - h + va;              wy = y - (VS - 1) + off + vb; }
            else if (dir == 0) { wx = x - off + (VS - 1 - vb); wy = y - h + va; }
            else if (dir == 1) { wx = x - h + (VS - 1 - va);   wy = y - off + (VS - 1 - vb); }
            else               { wx = x - VS + 1 + off + vb;   wy = y - h + (VS - 1 - va); }
        };
        for (int it = tid; it < n * VS * VS; it += kBlock) {
            const int k = it / (VS * VS), c = it - k * VS * VS, vb = c / VS, va = c - vb * VS;
            int wx, wy;
            world(s_rec[k], va, vb, wx, wy);
            uint32_t base = 0;
            if (wx >= 0 && wx < W && wy >= 0 && wy < H) base = g_grid[wx * H + wy];
            const bool transp = base == 0 || (cfg.obj[base].flags & MG_OF_SEE_BEHIND);
            if (transp) atomicOr(&s_trow[k * VS + vb], 1u << va);
        }
        __syncthreads();
        if (tid < n) {
            const uint64_t r = s_rec[tid];
            if (rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE) {          // base.py:743
                uint32_t m[MG_MAX_VIEW];
                if (cfg.see_through_walls) { for (int j = 0; j < VS; j++) m[j] = (1u << VS) - 1u; }
                else if (VS <= kRegView) occlude_rows<0>(VS, off, &s_trow[tid * VS], m);
                else occlude_rows_mem(VS, off, &s_trow[tid * VS], m);
                for (int vb = 0; vb < VS; vb++)
                    for (int va = 0; va < VS; va++)
                        if ((m[vb] >> va) & 1u) {
                            int wx, wy;
                            world(r, va, vb, wx, wy);
                            if (wx >= 0 && wx < W && wy >= 0 && wy < H) {     // (two cells share a dword: an atomic OR of the cell's half)
                                const int c = wx * H + wy;
                                atomicOr(reinterpret_cast<uint32_t*>(s_tile) + (c >> 1), kSeen << (16 * (c & 1)));
                            }
                        }
            }
        }
        __syncthreads();
    }
    const int tile_bytes = ts * ts * 3;
    // active 'prestige' agents: recoloured (and blended) tile per agent, world orientation (render_post)
    if (cfg.prestige_mask) {
        for (int X = 0; X < n; X++) {
            if (!((cfg.prestige_mask >> X) & 1u)) continue;
            const uint64_t rx = s_rec[X];
            if ((rec_byte(rx, MG_AG_FLAGS) & (MG_AF_ACTIVE | MG_AF_PLACED)) != (MG_AF_ACTIVE | MG_AF_PLACED)) continue;
            const uint32_t base = g_grid[rec_byte(rx, MG_AG_X) * H + rec_byte(rx, MG_AG_Y)];
            const uint32_t sdir = rec_byte(rx, MG_AG_DIR);
            const PrestigeColor col = prestige_color(st.prestige[(size_t)e * n + X], cfg.prestige_scale[X]);
            const uint32_t amax = (amax4 >> (8 * sdir)) & 0xFFu;
            const uint32_t M = ((amax * col.r) >> 8) + ((amax * col.g) >> 8) + ((amax * col.b) >> 8);
            const uint8_t* white = atlas + (size_t)(cfg.prestige_sprite_tile + sdir) * tile_bytes;   // no border
            const uint8_t* btile = base ? atlas + (size_t)(1 + base) * tile_bytes : nullptr;
            const bool border = base ? (cfg.obj[base].flags2 & 1) != 0 : true;
            const uint8_t* etile = atlas + (size_t)tile_bytes;
            for (int p = tid; p < ts * ts; p += kBlock)
                prestige_pixel(white[p * 3], col, M, btile ? btile + p * 3 : nullptr, border ? etile + p * 3 : nullptr,
                               s_dyn + (size_t)X * tile_bytes + p * 3);
        }
    }
    __syncthreads();
    // raster: image [H*ts][W*ts][3]; row R -> world y = R / ts; dword d of the row -> world x = d / TD
    const int TD = ts * 3 / 4;                    // dwords per tile row (ts % 4 == 0)
    const int row_dw = W * TD, tile_dw = ts * TD;
    const uint32_t* atlas32 = reinterpret_cast<const uint32_t*>(atlas);
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out + (size_t)blockIdx.x * H * ts * W * ts * 3);
    const int total = H * ts * row_dw;
    for (int q = tid; q < total; q += kBlock) {
        const int R = q / row_dw, d = q - R * row_dw;
        const int j = R / ts, rr = R - j * ts, i = d / TD, kk = d - i * TD;
        const int c = i * H + j;
        const uint32_t tc = s_tile[c], t = tc & (kSeen - 1u);
        uint32_t v = (t < n_tiles) ? atlas32[(size_t)t * tile_dw + rr * TD + kk]
                                   : reinterpret_cast<const uint32_t*>(s_dyn)[(size_t)(t - n_tiles) * tile_dw + rr * TD + kk];
        if (tc & kSeen) {
            // (img*8 + 255*2) >> 3, clipped to 255 (base.py:327-329) == min(255, img + 63) per byte
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t x = ((v >> (8 * b)) & 0xFFu) + 63u;
                o |= (x > 255u ? 255u : x) << (8 * b);
            }
            v = o;
        }
        o32[q] = v;
    }
}

hipError_t launch_frame(const MgConfig& cfg, const MgState& st, const int32_t* env_ids, int K, const uint8_t* atlas,
                        int ts, int highlight, uint32_t amax, uint8_t* out, hipStream_t s) {
    if (K <= 0) return hipSuccess;
    if ((size_t)cfg.n_tiles + (size_t)cfg.n_agents >= 0x8000u) return hipErrorInvalidValue;      // (bit 15 of a cell's tile index is its highlight)
    size_t lds = 2 * (size_t)round_up(cfg.W * cfg.H, 8) + MG_MAX_AGENTS * 8 +
                 (size_t)round_up(cfg.n_agents * cfg.view_size, 4) * 4 + 64 +
                 (cfg.prestige_mask ? (size_t)cfg.n_agents * ts * ts * 3 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(frame_kernel, dim3(K), dim3(kBlock), lds, s, cfg, st, env_ids, atlas, ts, highlight, amax, out);
    return hipGetLastError();
}

}  // namespace mg

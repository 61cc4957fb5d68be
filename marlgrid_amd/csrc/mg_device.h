// mg_device.h — device-side helpers shared by the gfx950 kernels of libmarlgrid_hip.so.
//
// Written for CDNA4 only (wave64, 160 KiB LDS/CU); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "marlgrid_hip.h"
#include "mg_core.h"

namespace mg {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // workgroup size of the lane-per-env / lane-per-cell kernels

// intra-wave LDS hand-off: DS operations of one wave execute in order, so only the compiler has
// to be kept from reordering across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// GridAgentInterface.render_post (marlgrid/agents.py:92-119): the sprite colour of an active
// 'prestige' agent, between red (prestige 0) and blue: (ps*blue + (1-ps)*red).astype(int)
struct PrestigeColor { uint32_t r, g, b; };
__device__ inline PrestigeColor prestige_color(double prestige, double scale) {
    const double ps = tanh(prestige / scale);
    PrestigeColor c;
    c.r = (uint32_t)(long long)(ps * 0.0 + (1. - ps) * 255.0);
    c.g = (uint32_t)(long long)(ps * 0.0 + (1. - ps) * 0.0);
    c.b = (uint32_t)(long long)(ps * 255.0 + (1. - ps) * 0.0);
    return c;
}

// One pixel of the tile a recoloured agent produces: alpha = the white sprite's coverage value,
// `base` = the overlappable object's pixel it stands on (blend_tiles, base.py:260-273) or NULL,
// `border` = the empty tile's pixel when the border rule applies (base.py:296-298) or NULL.
__device__ inline void prestige_pixel(uint32_t alpha, const PrestigeColor& col, uint32_t M, const uint8_t* base,
                                      const uint8_t* border, uint8_t* out) {
    uint32_t v[3] = {(alpha * col.r) >> 8, (alpha * col.g) >> 8, (alpha * col.b) >> 8};
    if (base) {
        if (M == 0) { v[0] = base[0]; v[1] = base[1]; v[2] = base[2]; }
        else {
            const uint32_t al = v[0] + v[1] + v[2];
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = ((uint32_t)base[c] * (M - al) + v[c] * al) / M;
        }
    }
    if (border) { v[0] += border[0]; v[1] += border[1]; v[2] += border[2]; }     // uint8 wrap-around add
    out[0] = (uint8_t)v[0]; out[1] = (uint8_t)v[1]; out[2] = (uint8_t)v[2];
}

// What mg_step_render adds to a launch of the obs kernel: the wave that renders an env first steps it.
struct FusedStep {
    const void* actions;      // [B][n], action_bytes each
    float* rewards;           // [B][n]
    int32_t action_bytes, enabled, has_prog;
    MgGenProgram prog;        // auto-reset program (has_prog)
    uint8_t* encode_out;      // mg_step_render_encode: MultiGrid.encode of the stepped batch, [B][W][H][3] (null: not asked for)
    uint32_t enc_m_cells, enc_m_n;   // ... its divide-by-multiply constants: ceil(2^32 / (W * H)), ceil(2^32 / n)
    int32_t enc_ne;           // ... dwords of its LDS table — one per grid byte value: object kinds, then the agent codes n_obj +
                              //     4 k + dir —, (n_obj + 4 n) rounded up to 16 (mg_render.hip: render_enc_entries; 0: none)
};

// Block-shared LDS of the obs-render kernel behind the atlas, sized by the configuration (object kinds in sixteens): per
// object kind its flags, overlap slot, flags2 and — with hide_item_types — the mask of the agents that hide it; per agent
// its prestige scale and the viewer map; for the fused step the object table (32 B per kind) and the first kOpsLds ops of
// the reset program (the rest, if any, is read in place).  Offsets in bytes from the end of the atlas.
constexpr int kOpsLds = 32;
struct RenderShared { int no, oflags, oslot, oflags2, hideby, pscale, vmap, obj, ops, total; };
__host__ __device__ inline RenderShared render_shared_layout(const MgConfig& cfg) {
    RenderShared h;
    h.no = ((cfg.n_obj < 1 ? 1 : cfg.n_obj) + 15) & ~15;
    int o = 0;
    h.oflags = o;  o += h.no;
    h.oslot = o;   o += h.no;
    h.oflags2 = o; o += h.no;
    h.hideby = o;  o += cfg.any_hide ? h.no * 4 : 0;          // uint32 [no]
    h.pscale = o;  o += MG_MAX_AGENTS * 8;                    // double [MG_MAX_AGENTS]
    h.vmap = o;    o += MG_MAX_AGENTS;                        // uint8 [MG_MAX_AGENTS]
    o = (o + 15) & ~15;
    h.obj = o;     o += h.no * 32;                            // MgObjDesc [no]
    h.ops = o;     o += kOpsLds * 32;                         // MgGenOp [kOpsLds]
    h.total = o;
    return h;
}

// x / d for small operands (x * d < 2^32) by multiply-high with ceil(2^32 / d): item index -> (slot, rest)
struct SmallDiv {
    uint32_t d, m;
    __host__ __device__ explicit SmallDiv(uint32_t d_) : d(d_), m(d_ > 1 ? 0xFFFFFFFFu / d_ + 1u : 0u) {}
    __device__ uint32_t div(uint32_t x) const { return d > 1 ? __umulhi(x, m) : x; }
};

// x / d with ONE full-rate multiply: on CDNA a 32-bit v_mul_lo / v_mul_hi issues at a quarter of the rate of the
// 24-bit v_mul_u32_u24 (and of every add, shift and compare), and the view / raster index arithmetic is made of
// divisions — which is why the instantiations off the HBM-bound fast path were VALU-bound.  m = ceil(2^20 / d);
// (x * m) >> 20 is x / d or x / d + 1 for every x < 2^20 whose quotient is < 2^11 (the product stays under 2^32,
// both operands under 2^24), and the one compare-and-subtract makes it exact.  `exact`: the caller knows
// x * (m * d - 2^20) < 2^20 for every x it passes (small compile-time divisors), so the fix-up is dropped.
struct Div20 {
    uint32_t d, m;
    __host__ __device__ explicit Div20(uint32_t d_) : d(d_), m(((1u << 20) + d_ - 1u) / d_) {}
    __host__ __device__ Div20(uint32_t d_, uint32_t m_) : d(d_), m(m_) {}      // (m worked out by the launcher)
    template <bool exact = false>
    __device__ __forceinline__ uint32_t div(uint32_t x) const {
        uint32_t q = __umul24(x, m) >> 20;
        if constexpr (!exact) q -= (__umul24(q, d) > x) ? 1u : 0u;
        return q;
    }
};

// per-wave LDS scratch of the obs-render kernel (bytes), shared by host launch code and kernel
struct RenderScratch {
    int grid, rec, pres, pcol, vaff, first, second, trow, trow2, tmap, dyn, out, step, total;
    int stage_envs;    // envs whose inputs (grid + agent records) are staged per batch: 1..8
    int tmap_slots;    // tmaps a wave can hold at once (= stage_envs): the look-ahead depth of its env loop
    int tmap_stride;   // bytes per tmap slot
    int rec_stride;    // u64 records per staged env
    int piece_rows;    // assemble-and-stream raster: pixel rows assembled in LDS per piece (0: chunk raster)
    int out_chunks;    // ... and the size of its piece buffer in 16-byte chunks
    int view_slots;    // envs whose views are derived together: slots of first / second / trow (1, or stage_envs)
    int cell_stride;   // bytes per slot of first / second
    int trow_stride;   // dwords per slot of trow
};
// The atlas in LDS.  As it is in HBM ([4 orientations][n_tiles][ts][ts][3], rounded up to 16 bytes) — except for the
// GATHER raster (mg_gather.h; the kernel's RM_ == 2, `mode` 2 below), which is instantiated for the reference's default
// view with its default 5-pixel tiles (agents.py:21-22), for 6-, 7-, 9-, 10-, 11- and 12-pixel tiles (the tile sizes
// under 16 that the 16-byte-chunk raster does not take) and for views 3, 5, 9 at 5-pixel tiles: there every tile ROW gets 16 zero bytes in
// front (GatherGeom::RS bytes per row, 32 zero bytes behind the last), so that a 16-byte window anywhere around a row is
// whole aligned dwords with zeros outside the row — no edge masks, no conditional reads.
__host__ __device__ inline bool render_gather(const MgConfig& cfg) {
    const int vs = cfg.view_size, ts = cfg.tile_size;
    // ('prestige' agents — per-env recoloured tiles next to the atlas's —: the reference's example, 11-pixel tiles, and the default 5)
    if (vs == 7) return (ts == 5 || ts == 6 || ts == 7 || ts == 9 || ts == 10 || ts == 11 || ts == 12) && (cfg.prestige_mask == 0 || ts == 11 || ts == 5);
    // the other view sizes the 16-byte-chunk raster is instantiated for (3 .. 9, even ones included) and the large odd views
    // (11, 13, 15: 8-wave workgroups — their shadow-cast arrays need more than 128 VGPRs), at GridAgentInterface's default tile size
    return (vs == 3 || vs == 4 || vs == 5 || vs == 6 || vs == 8 || vs == 9 || vs == 11 || vs == 13 || vs == 15) && ts == 5 && cfg.prestige_mask == 0;
}
__host__ __device__ inline int render_gather_row_bytes(int ts) { return (16 + 3 * ts + 3) / 4 * 4; }
__host__ __device__ inline int render_atlas_raw_bytes(const MgConfig& cfg) {
    return (4 * cfg.n_tiles * cfg.tile_size * cfg.tile_size * 3 + 15) / 16 * 16;
}
// `mode`: the kernel's RM_ (0: by tile size, 1: assemble-and-stream forced — measurement builds —, 2: gather, 3: assemble-and-
// stream with the grid AND the atlas read in place — grids that do not fit LDS)
__host__ __device__ inline int render_atlas_lds_bytes(const MgConfig& cfg, int mode) {
    if (mode == 3) return 0;
    if (mode == 2) return (4 * cfg.n_tiles * cfg.tile_size * render_gather_row_bytes(cfg.tile_size) + 32 + 15) / 16 * 16;
    return render_atlas_raw_bytes(cfg);
}

// What a launch of the obs kernel would otherwise work out in every wave before it requests its first byte — the
// LDS layout (render_scratch_for: four candidate layouts), the envs per wave, the dividers' multipliers: ~600
// instructions, 1.5 us of the launch's store-free head (tools/phase_stamps.py) — worked out by the launcher instead.
struct RenderLaunch {
    RenderScratch L;
    int per_wave;                               // envs per wave of the persistent grid
    uint32_t m_n, m_nv, m_nvVV, m_VV, m_VS, m_nvVS;   // Div20 multipliers of n, nv, nv * VS^2, VS^2, VS, nv * VS
    int depth_mode;                             // measurement builds: look-ahead depth forced for all waves (0: by wave)
    int atlas_lds;                              // bytes the atlas takes in LDS (render_atlas_lds_bytes; 0: read in place)
    RenderShared sh;                            // the block-shared tables behind it (render_shared_layout)
#if defined(MG_AB_VARIANTS)
    unsigned long long* stamps;                 // measurement build: phase stamps of every wave (tools/phase_stamps.py), or null
#endif
};
// n: the env's agents (records, who stands where); nv: the viewers this launch renders (view-sized arrays)
__host__ __device__ inline RenderScratch render_scratch_layout(int cells_stride, int n, int nv, int vs, int stage_envs = 1,
                                                               int dyn_bytes = 0, int out_bytes = 0, int piece_rows = 0,
                                                               bool any_hide = true, int max_view_slots = 0, bool gather = false,
                                                               bool big = false) {
    RenderScratch s;
    int o = 0;
    s.stage_envs = stage_envs;
    s.rec_stride = round_up(n * 8, 16) / 8;
    // (big: a grid too large for LDS — the kernel's RM_ == 3 — is read in place, and who stands on a view cell is searched among
    // the env's agents instead of looked up in per-cell maps: no `grid`, `first`, `second`)
    s.grid = o;  o += big ? 0 : stage_envs * round_up(cells_stride, 16);
    s.rec = o;   o += stage_envs * s.rec_stride * 8;
    s.pres = o;  o += dyn_bytes ? stage_envs * s.rec_stride * 8 : 0;   // agent.prestige of the staged envs
    s.pcol = o;  o += dyn_bytes ? round_up(stage_envs * s.rec_stride * 4, 16) : 0;   // ... and the sprite colours it gives them (fused step)
    // Views of a GROUP of envs at once: a slot of first (second: only with hide_item_types — the agents of a cell) and
    // of trow (transparency rows; visibility replaces them in place) per env of the group; a view cell's (object,
    // agent) pair waits in the env's tmap slot.
    // (chunk raster — out_bytes == 0 —: up to 4 envs of views at a time; it is HBM-bound and a wave's first store
    // should not wait for eight envs of views; the assemble-and-stream rasters take the whole staged batch)
    // (the gather raster: no piece buffer either, but a group's raster is ONE stream over all its envs: the whole batch)
    s.view_slots = out_bytes == 0 && !gather && stage_envs > 4 ? 4 : stage_envs;
    if (max_view_slots > 0 && s.view_slots > max_view_slots) s.view_slots = max_view_slots;
    s.cell_stride = round_up(cells_stride, 16);
    // (trow doubles as the per-agent colour words of the 'prestige' recolouring: at least n dwords)
    s.trow_stride = round_up((nv * vs > n ? nv * vs : n) * 4, 16) / 4;
    s.vaff = o;  o += round_up(s.view_slots * nv * 8, 16);   // per viewer: its view's affine map and identity (phase 2b)
    s.first = o; o += big ? 0 : s.view_slots * s.cell_stride;
    s.second = o; o += any_hide && !big ? s.view_slots * s.cell_stride : 0;
    s.trow = o;  o += s.view_slots * s.trow_stride * 4;
    // (views of more than 15 rows: the shadow cast walks its rows in memory — mg_occlude.h —, the result next to the transparency)
    s.trow2 = o; o += vs > 15 ? s.view_slots * s.trow_stride * 4 : 0;
    s.tmap_slots = stage_envs;
    s.tmap_stride = gather ? nv * vs * vs * 2 : round_up(nv * vs * vs * 2, 16);   // (gather: DENSE — band g of a group is entry g * vs)
    s.tmap = o;  o += round_up(s.tmap_slots * s.tmap_stride, 16);
    s.dyn = o;   o += round_up(dyn_bytes, 16);   // per-env recoloured ('prestige') agent tiles
    s.out = o;   o += round_up(out_bytes, 16);   // assemble-and-stream raster: the piece being assembled
    s.piece_rows = piece_rows;
    s.out_chunks = round_up(out_bytes, 16) / 16;
    // fused step (mg_step_render): lane j < stage_envs steps staged env j; its [item][8] columns: records, RNG look-ahead,
    // actions, the agent-parallel resolution's flags and turns (step_par_*, mg_core.h), the envs' step counts
    s.step = o;  o += round_up(n * 8 * 8 + MG_MT_HEAD * 8 * 4 + 3 * n * 8 + 8 * 4, 16);
    s.total = o;
    return s;
}
// the tile sizes the 16-byte-chunk raster is instantiated for (whole pairs of dwords per tile row);
// everything else — and `mode` 1, measurement builds — takes the assemble-and-stream raster
__host__ __device__ inline bool render_chunk_raster(const MgConfig& cfg, int mode) {
    return (cfg.tile_size == 8 || cfg.tile_size == 16 || cfg.tile_size == 32) && mode == 0;
}
// The layout a launch of the obs kernel uses, from the config and the workgroup size alone (kernel and
// launcher agree): recoloured-tile space when some agent is 'prestige'; for the assemble-and-stream raster
// (tile sizes off the 16-byte-chunk path; `mode` 1 forces it — measurement builds) the piece buffer: as
// many whole pixel rows as fit ~4 KiB (at least one) plus 32 bytes for the carried-over partial chunk;
// and as many staged envs per batch (8, 4, 2 or 1) as `wpb` waves of scratch leave room for.
__host__ __device__ inline RenderScratch render_scratch_for(const MgConfig& cfg, int wpb, int mode = 0) {
    const int n = cfg.n_agents, vs = cfg.view_size, ts = cfg.tile_size;
    const int nv = cfg.n_view ? cfg.n_view : n;
    // (the recoloured tiles of a 'prestige' env: as the atlas's — gather raster: in padded rows)
    const int dyn = cfg.prestige_mask ? (cfg.any_hide ? 2 : 1) * n * 4 * ts * (mode == 2 ? render_gather_row_bytes(ts) : ts * 3) + (mode == 2 ? 32 : 0) : 0;
    // (`fixed`: what a workgroup holds besides its waves' scratch — exactly the launcher's sum, launch_render_t)
    const int atlas_b = render_atlas_lds_bytes(cfg, mode), fixed = render_shared_layout(cfg).total;
    const bool gather = mode == 2, big = mode == 3;
    int rows = 0, out = 0;
    if (!gather && !render_chunk_raster(cfg, mode)) {
        const int rb = 3 * vs * ts;
        rows = 4096 / rb;
        if (rows < 1) rows = 1;
        if (rows > nv * vs * ts) rows = nv * vs * ts;
        out = 32 + rows * rb;
    }
    // the views of several envs are derived together — one lane per viewer in the shadow cast (its ~420 instructions
    // run once per group instead of once per env), full trips in the per-cell phases — with one slot of view scratch per
    // env of the group (see the kernel's pass 0); the recoloured tiles of a 'prestige' env keep their one slot (they are
    // made right before the env's raster)
    const RenderScratch b = render_scratch_layout(cfg.cells_stride, n, nv, vs, 1, dyn, out, rows, true, 0, gather, big);
    const int resident = (atlas_b + 4 * b.total + fixed <= 160 * 1024) ? atlas_b : 0;   // else the atlas is read in place
    // (a wave stages its batch's records with two per lane — mg_render_kernel.h, step_load_issue —: up to 8 envs of up to
    // 16 agents, 4 envs of more)
    const int kmax = n > 16 ? 4 : 8;
    // ('prestige' — 12-wave workgroups next to a large atlas —: fewer view slots before fewer staged envs or fewer waves)
    for (int slots = dyn ? kmax : 0; dyn && slots >= 1; slots >>= 1) {
        const RenderScratch t = render_scratch_layout(cfg.cells_stride, n, nv, vs, kmax, dyn, out, rows, cfg.any_hide != 0, slots, gather, big);
        if (resident + wpb * t.total + fixed <= 160 * 1024) return t;
    }
    int k = kmax;
    while (k > 1 && resident + wpb * render_scratch_layout(cfg.cells_stride, n, nv, vs, k, dyn, out, rows, cfg.any_hide != 0, dyn ? 1 : 0, gather, big).total + fixed > 160 * 1024) k >>= 1;
    return render_scratch_layout(cfg.cells_stride, n, nv, vs, k, dyn, out, rows, cfg.any_hide != 0, dyn ? 1 : 0, gather, big);
}

}  // namespace mg

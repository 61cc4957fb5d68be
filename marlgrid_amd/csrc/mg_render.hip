// mg_render.hip — the launcher of the observation raster (mg_render_obs / mg_step_render): which instantiation of
// mg::render_kernel (mg_render_kernel.h) a configuration gets.  The instantiations themselves are made, in parallel, by
// mg_render_inst_{a..e}.hip (and _v: the measurement variants); here they are only referred to.
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>   // getenv: the measurement build only (libmarlgrid_hip_ab.so, loaded by tools/)
#endif

namespace mg {

#if defined(MG_AB_VARIANTS)
unsigned long long* g_ab_stamps = nullptr;
extern "C" int mg_ab_stamps(unsigned long long* p) { g_ab_stamps = p; return 0; }
#endif

#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_A(MG_RENDER_EXTERN)
MG_RENDER_GROUP_N(MG_RENDER_EXTERN)
MG_RENDER_GROUP_B(MG_RENDER_EXTERN)
MG_RENDER_GROUP_C(MG_RENDER_EXTERN)
MG_RENDER_GROUP_D(MG_RENDER_EXTERN)
MG_RENDER_GROUP_E(MG_RENDER_EXTERN)
MG_RENDER_GROUP_G(MG_RENDER_EXTERN)
MG_RENDER_GROUP_X(MG_RENDER_EXTERN)
MG_RENDER_GROUP_H(MG_RENDER_EXTERN)
MG_RENDER_GROUP_I(MG_RENDER_EXTERN)
MG_RENDER_GROUP_J(MG_RENDER_EXTERN)
MG_RENDER_GROUP_K(MG_RENDER_EXTERN)
MG_RENDER_GROUP_L(MG_RENDER_EXTERN)
MG_RENDER_GROUP_M(MG_RENDER_EXTERN)
MG_RENDER_GROUP_V(MG_RENDER_EXTERN)
#endif

// The raster a configuration gets (the kernel's RM_): 2 = gather (mg_gather.h) where it is instantiated and its padded
// atlas fits LDS next to 4 waves of scratch, else 0 = by tile size (16-byte chunks / assemble-and-stream).
static int render_mode_for(const MgConfig& cfg) {
    if (!render_gather(cfg)) return 0;
    const RenderScratch L = render_scratch_for(cfg, 4, 2);
    return (size_t)render_atlas_lds_bytes(cfg, 2) + (size_t)render_shared_layout(cfg).total + 4 * (size_t)L.total <= 160 * 1024 ? 2 : 0;
}

// LDS of the smallest shape of the ordinary variants (4 waves, one staged env, the atlas read in place when it does not fit)
static int render_small_lds_bytes(const MgConfig& cfg) {
    const int mode = render_mode_for(cfg);
    const RenderScratch L = render_scratch_for(cfg, 4, mode);
    const int atlas_b = render_atlas_lds_bytes(cfg, mode);
    const int rest = render_shared_layout(cfg).total + 4 * L.total;
    return atlas_b + rest <= 160 * 1024 ? atlas_b + rest : rest;   // else the atlas is read in place
}
// A grid whose staged copy (and the per-cell first-agent maps beside it) does not fit LDS even then — beyond ~140 x 140, ~110 x
// 110 with hide_item_types — takes the variant that reads the grid in place (RM_ == 3; with 'prestige' agents: their recoloured tiles in LDS beside it).
static bool render_big_grid(const MgConfig& cfg) { return render_small_lds_bytes(cfg) > 160 * 1024; }

int render_min_lds_bytes(const MgConfig& cfg) {
    if (render_big_grid(cfg)) return render_shared_layout(cfg).total + 4 * render_scratch_for(cfg, 4, 3).total;
    return render_small_lds_bytes(cfg);
}

static size_t render_lds_bytes(const MgConfig& cfg, int wpb, int mode = 0) {
    const RenderScratch L = render_scratch_for(cfg, wpb, mode);
    return (size_t)render_atlas_lds_bytes(cfg, mode) + (size_t)render_shared_layout(cfg).total + (size_t)wpb * L.total;
}

// Workgroup shape.  16 waves per workgroup walk 16 *adjacent* envs at a time (a 450 KB contiguous
// output window per workgroup, one atlas copy per 16 waves): measured +7..13 % HBM write throughput
// over 4-wave workgroups at the bench batch.  Small batches keep 4-wave workgroups so that they
// still spread over all CUs.
static int choose_wpb(const MgConfig& cfg, int mode) {
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_RENDER_WPB")) { int w = atoi(f); if (w == 4 || w == 8 || w == 12 || w == 16) return w; }
#endif
#if defined(MG_EXP) && (MG_EXP & 6)      // experiment builds (tools/wpb_sweep.py): one workgroup shape for every batch
    return (MG_EXP & 6) == 2 ? 4 : (MG_EXP & 6) == 4 ? 8 : 16;
#endif
    return (cfg.B >= 4096 && render_lds_bytes(cfg, 16, mode) <= 160 * 1024) ? 16 : 4;
}

#define MG_RENDER_DISPATCH(VS, TS, V)                                                                      \
    (wpb == 16 ? launch_render_t<VS, TS, 16, V>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)           \
               : launch_render_t<VS, TS, 4, V>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick))
// run-time view size: its MG_MAX_VIEW-entry shadow-cast arrays need more than the 128 VGPRs a 16-wave
// workgroup leaves per lane (spills would be VMEM traffic in the middle of the run): 8-wave workgroups
#define MG_RENDER_DISPATCH_RT(TS, V)                                                                       \
    (wpb == 16 ? launch_render_t<0, TS, 8, V>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)             \
               : launch_render_t<0, TS, 4, V>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick))
// mg_step_render_encode: the instantiations with the encode compiled in (variant + 16, MG_RENDER_GROUP_N) — the shapes of the
// BASELINE configs (views 7 and 9 at 8-pixel tiles), any other view at 8-pixel tiles, GridAgentInterface's defaults (view 7
// at 5-pixel tiles) —, chosen as launch_render chooses among the plain ones.  Everything else (and what the fused encode
// cannot do: a grid read in place, object ids and agent marks that do not share a byte, 'prestige' agents, an atlas in
// global memory) is hipErrorNotSupported: nothing is launched, the C ABI answers MG_E_UNSUPPORTED, hosts call mg_step_render
// and mg_encode.
static int render_enc_entries(const MgConfig& cfg) { return cfg.n_obj + 4 * cfg.n_agents <= 256 ? ((cfg.n_obj + 4 * cfg.n_agents + 15) & ~15) : 0; }
static hipError_t launch_render_enc(const MgConfig& cfg, const MgState& st, uint8_t* obs, hipStream_t s, const FusedStep* fs0,
                                    RenderPick* pick) {
    FusedStep fse = *fs0;
    fse.enc_ne = render_enc_entries(cfg);
    const FusedStep* fs = &fse;
#if defined(MG_DEV_ONLY)
    return launch_render_t<MG_DEV_ONLY>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick);
#else
    if (cfg.prestige_mask || render_big_grid(cfg) || fse.enc_ne == 0) return hipErrorNotSupported;
    const int vs = cfg.view_size, ts = cfg.tile_size, mode = render_mode_for(cfg), wpb = choose_wpb(cfg, mode);
    const size_t enc_lds = (size_t)fse.enc_ne * 4;
    if (render_lds_bytes(cfg, 4, mode) + enc_lds > 160 * 1024) return hipErrorNotSupported;     // (incl. an atlas that stays in global memory)
    const bool w16 = wpb == 16 && render_lds_bytes(cfg, 16, mode) + enc_lds <= 160 * 1024;
    if (mode == 2) {
        if (vs == 7 && ts == 5)
            return w16 ? launch_render_t<7, 5, 16, 16, 2>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick)
                       : launch_render_t<7, 5, 4, 16, 2>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick);
        return hipErrorNotSupported;
    }
    if (ts != 8) return hipErrorNotSupported;
    if (vs == 7)
        return w16 ? launch_render_t<7, 8, 16, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick)
                   : launch_render_t<7, 8, 4, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick);
    if (vs == 9)
        return w16 ? launch_render_t<9, 8, 16, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick)
                   : launch_render_t<9, 8, 4, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick);
    // the views whose plain launch is a specialised instantiation the encode set does not have (3 ... 6, 8 at 8-pixel tiles): a
    // second launch costs them +7 %, the run-time-view instantiation would cost +40 %
    if (vs <= 9) return hipErrorNotSupported;
    // (run-time view size: 8-wave workgroups, as MG_RENDER_DISPATCH_RT)
    return wpb == 16 && render_lds_bytes(cfg, 8, mode) + enc_lds <= 160 * 1024
               ? launch_render_t<0, 8, 8, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick)
               : launch_render_t<0, 8, 4, 16, 0>(cfg, st, obs, nullptr, nullptr, nullptr, s, fs, pick);
#endif
}
bool render_can_encode(const MgConfig& cfg) {
    FusedStep f;
    f.enabled = 1;
    RenderPick p;
    return launch_render_enc(cfg, MgState{}, nullptr, nullptr, &f, &p) == hipSuccess;
}

// The kernel launch of mg_render_obs / mg_step_render.
hipError_t launch_render(const MgConfig& cfg, const MgState& st, uint8_t* obs, uint8_t* view_cells,
                         uint8_t* view_agent, uint8_t* vis_mask, hipStream_t s, const FusedStep* fs, RenderPick* pick) {
    if (cfg.B <= 0) return hipSuccess;

    FusedStep none;
    none.enabled = 0;
    none.has_prog = 0;
    none.actions = nullptr;
    none.rewards = nullptr;
    none.action_bytes = 8;
    none.prog.template_grid = nullptr;
    none.prog.n_ops = 0;
    none.prog.ops = nullptr;
    none.prog.reject = nullptr;
    none.prog.n_reject = 0;
    none.encode_out = nullptr;
    none.enc_m_cells = none.enc_m_n = 0;
    none.enc_ne = 0;
    if (!fs) fs = &none;
    if (fs->encode_out) {
        if (view_cells || view_agent || vis_mask || pick) return hipErrorInvalidValue;
        return launch_render_enc(cfg, st, obs, s, fs, nullptr);
    }
    if ((view_cells || view_agent || vis_mask) && !(view_cells && view_agent && vis_mask)) return hipErrorInvalidValue;
#if defined(MG_DEV_ONLY)   // development: compile ONE instantiation (register / ISA checks without the other sixty),
    // e.g. -DMG_DEV_ONLY="7,5,16,0,0"
    return launch_render_t<MG_DEV_ONLY>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
#else
    const int vs = cfg.view_size, ts = cfg.tile_size;
    if (render_big_grid(cfg))       // the grid read in place (everything about the view and the tiles at run time, 4-wave workgroups)
        return cfg.prestige_mask ? launch_render_t<0, 0, 4, 12, 3>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                                 : launch_render_t<0, 0, 4, 8, 3>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
    const int mode = render_mode_for(cfg);
    const int wpb = choose_wpb(cfg, mode);
    if (cfg.prestige_mask) {   // per-env recoloured agent tiles (LDS), 4-wave workgroups
        const RenderScratch L = render_scratch_for(cfg, 4);
        const size_t lds4 = (size_t)render_atlas_lds_bytes(cfg, 0) + (size_t)render_shared_layout(cfg).total +
                            4 * (size_t)L.total;
        if (lds4 > 160 * 1024) {   // the static atlas stays in global memory
            if (ts == 8) return launch_render_t<0, 8, 4, 12>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            if (ts == 16) return launch_render_t<0, 16, 4, 12>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            if (ts == 32) return launch_render_t<0, 32, 4, 12>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            return launch_render_t<0, 0, 4, 12>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        }
        // the shipped view: compile-time size; 8-wave workgroups when they fit (the recolouring code needs
        // more than the 128 VGPRs a 16-wave workgroup leaves per lane)
        // 12-wave workgroups (3 waves per SIMD, 168 VGPRs: the recolouring code needs ~165) where their scratch
        // fits next to the atlas: 0.52 -> 0.59 of 8 TB/s with three 'prestige' agents at tile 8, 0.24 -> 0.29 for the
        // reference's example (one agent, tile 11) against 8-wave workgroups (profiles/r03/ab_offpath*.jsonl)
        int pw = wpb;
        if (pw == 16) pw = render_lds_bytes(cfg, 12) <= 160 * 1024 ? 12 : 8;
        if (vs == 7 && ts == 8)
            return pw == 12 ? launch_render_t<7, 8, 12, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                 : pw == 8 ? launch_render_t<7, 8, 8, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                           : launch_render_t<7, 8, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        bool rt_ts = false;
#if defined(MG_AB_VARIANTS)
        if (const char* f = getenv("MG_RENDER_RT_TS")) rt_ts = atoi(f) != 0;
#endif
        if (mode == 2 && !rt_ts) {             // examples/human_player.py's view_tile_size 11: the gather raster
            int gw = wpb;
            if (gw == 16) gw = render_lds_bytes(cfg, 12, 2) <= 160 * 1024 ? 12 : render_lds_bytes(cfg, 8, 2) <= 160 * 1024 ? 8 : 4;
            if (ts == 5)                       // ... and GridAgentInterface's default tile size
                return gw == 12 ? launch_render_t<7, 5, 12, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                     : gw == 8 ? launch_render_t<7, 5, 8, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                               : launch_render_t<7, 5, 4, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            return gw == 12 ? launch_render_t<7, 11, 12, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                 : gw == 8 ? launch_render_t<7, 11, 8, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                           : launch_render_t<7, 11, 4, 9, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        }
        if (vs == 7 && (ts % 8) != 0)
            return pw == 12 ? launch_render_t<7, 0, 12, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                 : pw == 8 ? launch_render_t<7, 0, 8, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                           : launch_render_t<7, 0, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        if (ts == 8) return launch_render_t<0, 8, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        if (ts == 16) return launch_render_t<0, 16, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        if (ts == 32) return launch_render_t<0, 32, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        return launch_render_t<0, 0, 4, 9>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
    }
    if (mode == 2) {   // the gather raster: view 7 (GridAgentInterface's default, agents.py:21) with 5- .. 12-pixel tiles; views 3 / 5 / 9 at its default 5-pixel tiles
#define MG_RENDER_DISPATCH_G(VS, TS)                                                                                 \
    (wpb == 16 ? launch_render_t<VS, TS, 16, 0, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)             \
               : launch_render_t<VS, TS, 4, 0, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick))
        if (vs > 9) {      // views 11 / 13 / 15: 8-wave workgroups where their scratch fits, else 4
            const int w8 = (cfg.B >= 4096 && render_lds_bytes(cfg, 8, 2) <= 160 * 1024) ? 8 : 4;
#define MG_RENDER_DISPATCH_G8(VS)                                                                                    \
    (w8 == 8 ? launch_render_t<VS, 5, 8, 0, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)                 \
             : launch_render_t<VS, 5, 4, 0, 2>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick))
            if (vs == 11) return MG_RENDER_DISPATCH_G8(11);
            if (vs == 13) return MG_RENDER_DISPATCH_G8(13);
            return MG_RENDER_DISPATCH_G8(15);
#undef MG_RENDER_DISPATCH_G8
        }
        if (vs == 3) return MG_RENDER_DISPATCH_G(3, 5);
        if (vs == 4) return MG_RENDER_DISPATCH_G(4, 5);
        if (vs == 5) return MG_RENDER_DISPATCH_G(5, 5);
        if (vs == 6) return MG_RENDER_DISPATCH_G(6, 5);
        if (vs == 8) return MG_RENDER_DISPATCH_G(8, 5);
        if (vs == 9) return MG_RENDER_DISPATCH_G(9, 5);
        switch (ts) {
        case 5: return MG_RENDER_DISPATCH_G(7, 5);
        case 6: return MG_RENDER_DISPATCH_G(7, 6);
        case 7: return MG_RENDER_DISPATCH_G(7, 7);
        case 9: return MG_RENDER_DISPATCH_G(7, 9);
        case 10: return MG_RENDER_DISPATCH_G(7, 10);
        case 11: return MG_RENDER_DISPATCH_G(7, 11);
        default: return MG_RENDER_DISPATCH_G(7, 12);
        }
#undef MG_RENDER_DISPATCH_G
    }
    {   // atlas too large for LDS (next to 4 waves of scratch): read it from global memory instead
        const RenderScratch L = render_scratch_for(cfg, 4);
        const size_t lds4 = (size_t)render_atlas_lds_bytes(cfg, 0) + (size_t)render_shared_layout(cfg).total +
                            4 * (size_t)L.total;
        if (lds4 > 160 * 1024) {
            if (ts == 8) return launch_render_t<0, 8, 4, 8>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            if (ts == 16) return launch_render_t<0, 16, 4, 8>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            if (ts == 32) return launch_render_t<0, 32, 4, 8>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
            return launch_render_t<0, 0, 4, 8>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
        }
    }
    if (ts == 8 && vs == 7) {
#if defined(MG_AB_VARIANTS)
        switch (getenv("MG_RENDER_VARIANT") ? atoi(getenv("MG_RENDER_VARIANT")) : 0) {   // tools/ab_render.py
        case 2: return MG_RENDER_DISPATCH(7, 8, 2);
        case 3: return MG_RENDER_DISPATCH(7, 8, 3);
        case 4: return MG_RENDER_DISPATCH(7, 8, 4);
        case 6: return MG_RENDER_DISPATCH(7, 8, 6);
        case 11: return MG_RENDER_DISPATCH(7, 8, 11);
        default: break;
        }
        if (getenv("MG_RENDER_RASTER") && atoi(getenv("MG_RENDER_RASTER")) == 1)   // assemble-and-stream at tile 8
            return wpb == 16 ? launch_render_t<7, 8, 16, 0, 1>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick)
                             : launch_render_t<7, 8, 4, 0, 1>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
#endif
#if defined(MG_EXP) && (MG_EXP & 8)      // experiment build: 12-wave workgroups (3 waves per SIMD, a 168-VGPR budget) for the plain kernel
        if (wpb == 16) return launch_render_t<7, 8, 12, 0, 0>(cfg, st, obs, view_cells, view_agent, vis_mask, s, fs, pick);
#endif
        return MG_RENDER_DISPATCH(7, 8, 0);
    }
    if (ts == 8 && vs == 9) return MG_RENDER_DISPATCH(9, 8, 0);
    if (ts == 8 && vs == 5) return MG_RENDER_DISPATCH(5, 8, 0);
    if (ts == 8 && vs == 3) return MG_RENDER_DISPATCH(3, 8, 0);
    if (ts == 8 && vs == 4) return MG_RENDER_DISPATCH(4, 8, 0);      // even views (agents.py:233-266 as it is written)
    if (ts == 8 && vs == 6) return MG_RENDER_DISPATCH(6, 8, 0);
    if (ts == 8 && vs == 8) return MG_RENDER_DISPATCH(8, 8, 0);
    if (ts == 8) return MG_RENDER_DISPATCH_RT(8, 0);      // other view sizes: run-time VS, same raster
    if (ts == 16 && vs == 7) return MG_RENDER_DISPATCH(7, 16, 0);
    if (ts == 32 && vs == 7) return MG_RENDER_DISPATCH(7, 32, 0);
    if (ts == 16) return MG_RENDER_DISPATCH_RT(16, 0);
    if (ts == 32) return MG_RENDER_DISPATCH_RT(32, 0);
    if (vs == 7) return MG_RENDER_DISPATCH(7, 0, 0);        // the default view with any other tile size (or a gather atlas too large for LDS)
    // the other small views with any tile size: compile-time view (exact dividers, a shadow cast of VS rows, 16-wave workgroups),
    // run-time tile size
    if (vs == 3) return MG_RENDER_DISPATCH(3, 0, 0);
    if (vs == 4) return MG_RENDER_DISPATCH(4, 0, 0);
    if (vs == 5) return MG_RENDER_DISPATCH(5, 0, 0);
    if (vs == 6) return MG_RENDER_DISPATCH(6, 0, 0);
    if (vs == 8) return MG_RENDER_DISPATCH(8, 0, 0);
    if (vs == 9) return MG_RENDER_DISPATCH(9, 0, 0);
    return MG_RENDER_DISPATCH_RT(0, 0);                   // anything else
#endif
}

}  // namespace mg

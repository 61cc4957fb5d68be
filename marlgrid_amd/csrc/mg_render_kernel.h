// mg_render_kernel.h — the observation raster: batched MultiGridEnv.gen_obs (marlgrid/base.py:418-474)
//   = get_view_exts + MultiGrid.slice + rotate_grid   (agents.py:237-266, base.py:123-147, 67-80)
//   + MultiGrid.opacity + occlude_mask                 (base.py:103-106, agents.py:290-343)
//   + MultiGrid.render / render_tile / blend_tiles     (base.py:275-331)
// producing obs[B][n][P][P][3] uint8.  This is the HBM-write-bound kernel of the engine: 9 408 B
// (view 7, tile 8) per agent-step out, ~60 B in.
//
// Mapping (CDNA4): ONE WAVEFRONT PER ENV, 16 waves per workgroup (4 for small batches), persistent
// grid; every wave walks its own contiguous run of envs.
//   * the whole pre-rotated sprite atlas ([4 orientations][n_tiles][ts*ts*3] bytes, ~21 KB for the
//     3-agent configs) is staged once per workgroup in LDS (read in place from L2 if it cannot fit);
//   * a BATCH of up to 8 envs at a time, the wave stages the envs' grids (W*H bytes each) and agent records in
//     LDS — in mg_step_render together with the env step's own inputs, and lanes 0..7 then STEP the staged envs
//     (mg_core.h step_run, auto-reset included) before anything is drawn —, derives the view_size x view_size
//     egocentric neighbourhoods (base object, shown agent, transparency) of a group of envs cooperatively, one
//     lane per viewer runs the shadow-casting pass as row bit-masks (log-step floods), and the wave writes a
//     per-view-cell atlas offset map (tmap) per env to LDS — the tile of a cell is selected where the cell is looked at,
//     the shadow cast's result applied by view row;
//   * the raster then emits the env's n*P*P*3 contiguous output bytes as 16-byte
//     (global_store_dwordx4) chunks, consecutive lanes -> consecutive chunks.  Tile sizes that are a
//     multiple of 8: each chunk is assembled in registers from two 8-byte LDS look-ups
//     atlas[tmap[cell] + row*TD + k].  The reference's default view with 5- (its default), 6-, 7-, 9- … 12-pixel tiles: the GATHER raster
//     (mg_gather.h) — a lane composes an aligned 16-byte chunk from the (at most two) tile rows it spans, which sit
//     in LDS padded with zeros, with aligned dword reads and v_alignbyte, and stores it.  Any other tile size: a few
//     KiB of whole pixel rows at a time are first ASSEMBLED in an LDS piece buffer — one lane per (row, view column)
//     segment ORs its 3*TS bytes from the atlas tile row into the zeroed buffer (aligned dwords cut with v_alignbyte
//     and ds_or_b32: unaligned DS accesses are serialised on gfx950) — and then STREAMED out as linear
//     ds_read_b128 -> aligned dwordx4 stores; the bytes of a chunk that straddles two pieces (or two
//     envs of the wave's run) are carried over in the buffer, so everything but the first and last
//     <16 bytes of a wave's whole run leaves as aligned 16-byte stores.
// No MFMA: there is no contraction anywhere in this path.
#pragma once
#include "mg_device.h"
#include "mg_launch.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>   // getenv: the measurement build only (libmarlgrid_hip_ab.so, loaded by tools/)
#endif
#include "mg_occlude.h"
#include "mg_gather.h"
#include "mg_encode_core.h"

namespace mg {

// ---- one (pixel row, view column) segment of the assemble-and-stream raster ------------------------
// SEG bytes from sb[sa ...] (any alignment) into the piece buffer at byte address `dl` (any alignment) with
// ALIGNED dword accesses only — an unaligned DS access is serialised lane by lane on gfx950 (measured: 15 x
// slower), and single-byte edge copies cost more instructions than the copy itself.  The buffer is zero
// before a piece is assembled, so a segment simply ORs (ds_or_b32) every dword it touches: the source is
// read as aligned dwords and cut with v_alignbyte to the destination's phase, the first and last dword are
// masked to the segment's own bytes — neighbouring segments, handled by other lanes, OR theirs into the
// same dwords.  All reads of a step are issued before its writes.  SEGC: SEG when the tile size is a
// compile-time constant (the loop unrolls), else 0.
template <int SEGC>
__device__ __forceinline__ void or_segment(const uint8_t* __restrict__ sb, uint32_t sa, uint8_t* __restrict__ lds0,
                                           uint32_t dl, uint32_t SEG) {
    const uint32_t phi = dl & 3u, nc = (phi + SEG + 3u) >> 2;            // destination dwords touched
    const int so = (int)sa - (int)phi;                                   // source byte that lands on byte 0 of dword 0
    const int a0 = so >> 2;                                              // (-1 when so < 0: that word is masked out anyway)
    const uint32_t sh = (uint32_t)so & 3u;
    const uint32_t* A = reinterpret_cast<const uint32_t*>(sb);
    uint32_t* D = reinterpret_cast<uint32_t*>(lds0 + (dl & ~3u));
    const uint32_t end = (phi + SEG) & 3u;
    const uint32_t m_first = 0xFFFFFFFFu << (8u * phi), m_last = end ? 0xFFFFFFFFu >> (8u * (4u - end)) : 0xFFFFFFFFu;
    constexpr int CH = SEGC ? ((SEGC + 6) / 4 < 10 ? (SEGC + 6) / 4 : 10) : 8;   // (a 33-byte segment — tile 11 — in one pass)
    uint32_t lo = a0 >= 0 ? A[a0] : 0u;
    for (uint32_t i0 = 0; i0 < nc; i0 += CH) {
        uint32_t w[CH + 1];
        w[0] = lo;
#pragma unroll
        for (int i = 0; i < CH; i++) w[i + 1] = (i0 + i < nc) ? A[a0 + (int)(i0 + i) + 1] : 0u;
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (i0 + i < nc) {
                uint32_t v = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
                if (i0 + i == 0) v &= m_first;
                if (i0 + i == nc - 1u) v &= m_last;
                atomicOr(&D[i0 + i], v);
            }
        }
        lo = w[CH];
    }
}

// A second look at a by-value kernel parameter, through a pointer the compiler cannot see through: the loads
// stay where the values are used (scalar loads from the kernarg segment) instead of being hoisted to the kernel's
// entry and carried — spilled — across the whole env loop.  (The parameter list of render_kernel as a struct: the
// kernarg segment is laid out by the same rules.)
struct RenderKernargs { MgConfig cfg; MgState st; uint8_t* obs; uint8_t* dbg[3]; RenderLaunch lc; FusedStep fs; };
template <class T>
__device__ __forceinline__ const T& kernarg_again(size_t offset) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T*)(p + offset);
}

// The launch constants of a REGION of the env loop (staging, step, an env's views and raster), re-read through
// kernarg_again and re-derived there: declared once in front of the loop they were ~290 SGPRs of loop invariants,
// spilled to VGPR lanes (five VGPRs of an instantiation that runs at its 128-VGPR limit) and read back with a
// v_readlane wherever one was used.  The names shadow the kernel's own.
#define MG_REGION_LOCALS                                                                                       \
    const MgConfig& cfg = kernarg_again<MgConfig>(offsetof(RenderKernargs, cfg));                              \
    const RenderLaunch& lc = kernarg_again<RenderLaunch>(offsetof(RenderKernargs, lc));                        \
    const RenderScratch& L = lc.L;                                                                             \
    uint8_t* const ws = smem + (kGlobalAtlas ? 0 : lc.atlas_lds) + lc.sh.total + (size_t)wave * L.total;       \
    uint8_t* const w_stage_g = ws + L.grid;                                                                    \
    uint64_t* const w_stage_r = reinterpret_cast<uint64_t*>(ws + L.rec);                                       \
    double* const w_stage_p = reinterpret_cast<double*>(ws + L.pres);                                          \
    uint32_t* const w_stage_c = reinterpret_cast<uint32_t*>(ws + L.pcol);                                      \
    uint2* const w_vaff = reinterpret_cast<uint2*>(ws + L.vaff);                                               \
    uint8_t* const w_first = ws + L.first;                                                                     \
    uint8_t* const w_second = ws + L.second;                                                                   \
    uint32_t* const w_trow = reinterpret_cast<uint32_t*>(ws + L.trow);                                         \
    uint32_t* const w_vis = w_trow;              /* (visibility replaces transparency in place) */              \
    uint16_t* const w_tmap0 = reinterpret_cast<uint16_t*>(ws + L.tmap);                                        \
    uint8_t* const w_dyn = ws + L.dyn;                                                                         \
    uint8_t* const w_out = ws + L.out;                                                                         \
    const uint32_t dyn_off = (uint32_t)(w_dyn - smem);                                                         \
    const int VS = VS_ ? VS_ : cfg.view_size, TS = TS_ ? TS_ : cfg.tile_size, VV = VS * VS, h = VS / 2;        \
    const int tile_bytes = TS * TS * 3;                                                                        \
    const size_t img_bytes = (size_t)VS * TS * VS * TS * 3;                                                    \
    const Div20 by_VV = VS_ ? Div20((uint32_t)VV) : Div20((uint32_t)VV, lc.m_VV);                              \
    const Div20 by_VS = VS_ ? Div20((uint32_t)VS) : Div20((uint32_t)VS, lc.m_VS);                              \
    (void)h; (void)tile_bytes; (void)img_bytes; (void)by_VV; (void)by_VS;                                      \
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H, nv = cfg.n_view ? cfg.n_view : n;                        \
    const int gdw = cfg.cells_stride / 4, off = cfg.view_offset, rec_stride = L.rec_stride;                    \
    const uint32_t NT4 = 4u * (uint32_t)cfg.n_tiles;                                                           \
    const Div20 by_n((uint32_t)n, lc.m_n), by_nv((uint32_t)nv, lc.m_nv), by_nvVV((uint32_t)(nv * VV), lc.m_nvVV); \
    (void)W; (void)H; (void)gdw; (void)off; (void)rec_stride; (void)NT4; (void)by_n; (void)by_nv; (void)by_nvVV;   \
    (void)w_stage_p; (void)w_stage_c; (void)w_vaff; (void)w_first; (void)w_second;                                 \
    (void)w_trow; (void)w_vis; (void)w_tmap0; (void)w_out; (void)dyn_off; (void)w_stage_g; (void)w_stage_r; (void)cfg

// ---- mg_step_render: the step of a batch of staged envs, lane j < kb steps env eb + j (mg_core.h) ----
// The wave steps the envs it is about to render ON THEIR STAGED COPIES: the loads whose addresses are known up
// front (records, actions, RNG look-ahead, counters: step_load) are issued together with the batch's grid loads
// — one HBM round trip — and every grid look-up of the action loop, of a respawn and of a fused reset is an
// LDS access.  What the step changes goes back to HBM with stores nobody waits for: records and counters from
// step_run, and the grid slices it reports as written (a pickup / drop / toggle, or a reset) from the wave.
// The views and the raster then read the stepped state where it already is.
// (Inlined: as a call, the by-value launch structs would be copied to per-lane scratch.  The 16-wave
// workgroups run at the 128-VGPR limit; the few dwords the step's live ranges spill are spilled and
// reloaded around this region, once per batch — checked in the ISA: no scratch access in the raster loops.)
__device__ __forceinline__ StepScratch fused_step_scratch(const MgConfig& cfg, int lane, uint8_t* sp, const MgObjDesc* s_obj,
                                                          const uint8_t* s_oflags) {
    const int n = cfg.n_agents;
    StepScratch sc;
    sc.rec = reinterpret_cast<uint64_t*>(sp);                                   // [n][8]
    sc.head = reinterpret_cast<uint32_t*>(sp + n * 8 * 8);                      // [MG_MT_HEAD][8]
    sc.act = sp + n * 8 * 8 + MG_MT_HEAD * 8 * 4;                               // [n][8]
    sc.pflag = sc.act + n * 8;                                                  // [n][8]  step_par_*: moved / needs the loop
    sc.ordp = sc.pflag + n * 8;                                                 // [n][8]  ... the agent's turn
    sc.ord = sc.ordp;                                                           // (more than 16 agents — no lane-parallel resolution —: iter_order)
    sc.psc = reinterpret_cast<int32_t*>(sc.ordp + n * 8);                       // [8]     ... the env's step count
    sc.rec_out = sc.rec;                                                        // (in place: the lanes of a wave run in lockstep)
    sc.fb = nullptr;                                                            // (the grid is a staged LDS copy: no pre-load)
    sc.obj = s_obj;
    sc.oflags = s_oflags;
    sc.S = 8;
    sc.col = lane;
    return sc;
}

// step_load (mg_core.h) for the batch of a wave, with all 64 lanes: in step_load lane j reads env j's records,
// actions and 16 RNG look-ahead words one after the other — 8 active lanes, every load instruction touching 8
// different cache lines, ~25 instructions: 3 us of the launch's store-free head.  The batch's records, actions
// and look-ahead words are each one CONTIGUOUS run in HBM, so the wave reads them as such (lane l: element l,
// l + 64) and transposes into the step's [item][8] LDS columns.  Same result as step_load on lanes 0 .. kb-1.
// In two halves — the loads (step_load_issue) and, after whatever the caller has to do in between, the wait and
// the LDS columns (step_load_commit; its `kb` may be smaller than the one the loads were issued for).
struct StepLoadRegs { uint64_t rv[2]; uint64_t a8[2]; uint32_t a4[2]; uint32_t a1[2]; uint32_t hv[2]; int pos0, sc0; };   // (a8 / a4 / a1: the action's raw bytes, by width)
__device__ __forceinline__ StepLoadRegs step_load_issue(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes,
                                                        int eb, int kb, int lane) {
    const int n = cfg.n_agents, nr = kb * n, nh = kb * MG_MT_HEAD;
    const uint64_t* rsrc = st.agents + (size_t)eb * n;
    const uint32_t* hsrc = st.mt_head + (size_t)eb * MG_MT_HEAD;
    StepLoadRegs r = {{0ull, 0ull}, {0ull, 0ull}, {0u, 0u}, {0u, 0u}, {0u, 0u}, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int i = lane + q * kWave;
        if (i < nr) r.rv[q] = rsrc[i];
        if (i < nh) r.hv[q] = hsrc[i];
    }
    if (lane < kb) { r.pos0 = st.mt_pos[eb + lane]; r.sc0 = st.step_count[eb + lane]; }
    // The actions LAST, raw, a register per width: anything done to a loaded value — a sign extension, even the
    // move that merges two widths into one variable — is a wait for everything requested so far, and hipcc waits
    // where the three widths' branches meet in any case: at the end of the requests that wait is the round trip's own.
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int i = lane + q * kWave;
        // (per-lane conditions — `lane` is opaque here —: as uniform branches the three widths are a diamond whose arms
        // zero the other widths' registers, and hipcc waits for every outstanding load before such a write)
        const size_t a = (size_t)eb * n + i;
        const bool live = i < nr && lane >= 0;
        if (live && action_bytes == 8) r.a8[q] = static_cast<const uint64_t*>(actions)[a];
        if (live && action_bytes == 4) r.a4[q] = static_cast<const uint32_t*>(actions)[a];
        if (live && action_bytes == 1) r.a1[q] = static_cast<const uint8_t*>(actions)[a];
    }
    return r;
}
__device__ __forceinline__ StepEnv step_load_commit(const MgConfig& cfg, const StepLoadRegs& r, int kb, int lane, const StepScratch& sc,
                                                    const Div20& by_n) {
    const int n = cfg.n_agents, nr = kb * n, nh = kb * MG_MT_HEAD;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int i = lane + q * kWave;
        if (i < nr) {
            const int j = (int)by_n.div((uint32_t)i), k = i - __mul24(j, n);
            sc.rec[k * 8 + j] = r.rv[q];
            // valid actions are 0..6 (step_load): as unsigned raw bytes of any width, exactly the values <= 6
            const uint64_t av = r.a8[q] | (uint64_t)(r.a4[q] | r.a1[q]);     // (only the launch's width was loaded; the others are 0)
            sc.act[k * 8 + j] = av <= 6ull ? (uint8_t)av : (uint8_t)0xFF;
        }
        if (i < nh) sc.head[(i % MG_MT_HEAD) * 8 + i / MG_MT_HEAD] = r.hv[q];
    }
    StepEnv e = {r.pos0, r.sc0};
    return e;
}

#if defined(MG_AB_VARIANTS)
// measurement build: wall_clock64 (100 MHz) of every wave's lane 0 at the phase boundaries of its FIRST batch —
// 0 entry, 1 tables + atlas in LDS, 2 batch staged (and step_load done), 3 batch stepped, 4 first views, 5 first
// env rastered, 6 wave done; 7: XCC_ID << 16 | HW_ID; 8..12 inside step_run of the first batch (StepScratch::stamp:
// 8 entry, 9 spawns + front cells + shuffle done, 10 agent loop done, 11 done / respawn / reset done, 12 state written
// back), 16..20 the same of the wave's LAST batch; 13 the launch's constants set up, 14 env loop entered, 15 the
// prologue's data is there — 24 words per wave, read by tools/phase_stamps.py
// (a kernel ARGUMENT of the measurement build — lc.stamps — so that the instantiations can live in several translation units)
extern unsigned long long* g_ab_stamps;       // set through mg_ab_stamps (mg_render.hip)
#define d_ab_stamps (lc.stamps)
#define MG_STAMP(slot) do { if (d_ab_stamps && lane == 0) d_ab_stamps[(size_t)(blockIdx.x * WPB + wave) * 24 + (slot)] = wall_clock64(); } while (0)
#else
#define MG_STAMP(slot) do {} while (0)
#endif

// ---- the kernel ----------------------------------------------------------------------------------
// TS_ % 8 == 0: 16-byte-chunk fast raster (tile rows are an even number of dwords); VS_ > 0 also
//              fixes the view size at compile time (the shipped view sizes), VS_ == 0 reads it from cfg.
// otherwise (or RM_ == 1): the assemble-and-stream raster; TS_ == 0 reads the tile size from cfg.
// V_: 0 = production; 8 = production with the atlas read from global memory (chosen by the launcher
//     when it does not fit LDS); 9 = production with per-env recoloured tiles for 'prestige' agents;
//     12 = both (recoloured tiles in LDS, the static atlas in global memory).  2..7 = measurement variants for tools/ab_render.py (MG_RENDER_VARIANT,
//     <7,8> only): 2 nontemporal stores, 3 raster only (phases 2-5 skipped), 4 stores only (no LDS
//     look-ups), 6 no store bursts, 11 phases 2-5 executed twice.
// WPB = waves per workgroup (4 or 16; MG_RENDER_WPB overrides the launcher's choice).
// VX_ = V_ + 16: mg_step_render_encode's instantiations — the fused step also writes MultiGrid.encode of its batch (compiled
//     in, not a flag of the launch: with the code in every instantiation the launches that do not ask for it lost 0.4 - 1.4 %
//     off the fast path — `profiles/r06/ab_fused_encode_in_launch_as_a_flag_v23.txt`).
template <int VS_, int TS_, int WPB, int VX_ = 0, int RM_ = 0>
__global__ __launch_bounds__(WPB * 64) void render_kernel(MgConfig cfg, MgState st, uint8_t* __restrict__ obs,
                                                        uint8_t* __restrict__ dbg_cells,
                                                        uint8_t* __restrict__ dbg_agent,
                                                        uint8_t* __restrict__ dbg_vis, RenderLaunch lc, FusedStep fs) {
    constexpr int V_ = VX_ & 15;                // the variant proper
    constexpr bool kEnc = (VX_ & 16) != 0;      // + 16: mg_step_render_encode
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // the wave index is uniform: told to the compiler, everything derived from it (the wave's scratch
    // pointers, its run of envs, loop bounds) lives in SGPRs instead of one VGPR each
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    MG_STAMP(0);
#if defined(MG_AB_VARIANTS)
    if (d_ab_stamps && lane == 0)      // where this wave runs: XCC_ID (hwreg 20) << 16 | HW_ID (hwreg 4) [15:0]
        d_ab_stamps[(size_t)(blockIdx.x * WPB + wave) * 24 + 7] =
            ((unsigned long long)(__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) & 0xF) << 16) |
            (unsigned long long)(__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4) & 0xFFFF);
#endif

    // Each wave walks its own CONTIGUOUS run of envs, i.e. one long sequential output stream per wave
    // (measured +5 % HBM write throughput over a grid-strided walk).
    constexpr bool kGlobalAtlas = (V_ == 8 || V_ == 12);
    const int atlas_bytes = kGlobalAtlas ? 0 : lc.atlas_lds;       // in LDS (render_atlas_lds_bytes)
    // RM_ == 2: the gather raster (mg_gather.h) — tile rows padded with zeros in LDS: 16 zero bytes in front of every row
    constexpr bool kGather = RM_ == 2;
    static_assert(!kGather || (VS_ > 0 && TS_ >= 5 && (TS_ % 8) != 0 && (V_ == 0 || V_ == 9)), "gather raster: compile-time view and tile size, static atlas in LDS");
    typedef GatherGeom<kGather ? VS_ : 7, kGather ? TS_ : 5> Gm;
    constexpr bool kPadRows = kGather;                                // (the prologue pads the atlas as it copies it)
    // RM_ == 3: a grid that does not fit LDS (beyond ~140 x 140) is read IN PLACE — the step works on its home in HBM, a view
    // cell's object is a global load, and who stands on a view cell is searched among the env's agents instead of looked up in
    // per-cell maps; the atlas is read in place too (V_ == 8), view and tile size are run-time values
    constexpr bool kBigGrid = RM_ == 3;
    static_assert(!kEnc || !kBigGrid, "mg_step_render_encode works on the staged grids");
    static_assert(!kBigGrid || (VS_ == 0 && TS_ == 0 && (V_ == 8 || V_ == 12)), "grid read in place: the fully run-time instantiations");
    constexpr int kRowB = kGather ? Gm::RS : 0, kRowW = kRowB / 4;
    constexpr int kPadFrontW = Gm::FRONT / 4, kPadTailW = Gm::TAIL / 4;   // zero dwords in front of a row / behind the last
    constexpr int kPadQ = 6;                                         // padded dwords per thread in the prologue's first round trip
    const int per_wave = lc.per_wave, depth_mode = lc.depth_mode;
    const int e0 = (blockIdx.x * WPB + wave) * per_wave;
    const int e_end = min(cfg.B, e0 + per_wave);
    constexpr int kSR = 8;                                              // grid dwords per lane and round trip

    // ---- block-shared: atlas + object flags ----
    // V_ == 8: the atlas does not fit the 160 KiB of LDS next to the per-env scratch (large tiles);
    // it is then read in place (global memory, L2-resident: it is a few hundred KB).
    constexpr bool kPrestige = (V_ == 9 || V_ == 12);   // some agent is 'prestige'-coloured: per-env recoloured tiles
    constexpr bool kSplit = (V_ == 12);        // static tiles in global memory, recoloured ones in LDS
    constexpr uint32_t kInLds = 0x80000000u;   // kSplit: marks a source offset as relative to the LDS base
    uint8_t* s_atlas = smem;
    // the block-shared tables (render_shared_layout: sized by the configuration's object kinds, in sixteens)
    uint8_t* const s_shared = smem + atlas_bytes;
    const int NO = lc.sh.no;                                                            // object kinds, padded
    uint8_t* s_oflags = s_shared + lc.sh.oflags;                                        // [NO]
    uint8_t* s_oslot = s_shared + lc.sh.oslot;                                          // [NO]
    uint8_t* s_oflags2 = s_shared + lc.sh.oflags2;                                      // [NO]
    uint32_t* s_hideby = reinterpret_cast<uint32_t*>(s_shared + lc.sh.hideby);          // [NO] who hides the kind (any_hide only)
    double* s_pscale = reinterpret_cast<double*>(s_shared + lc.sh.pscale);              // [MG_MAX_AGENTS] prestige_scale
    uint8_t* s_vmap = s_shared + lc.sh.vmap;                                            // [MG_MAX_AGENTS] viewer slot -> agent
    MgObjDesc* s_obj = reinterpret_cast<MgObjDesc*>(s_shared + lc.sh.obj);              // [NO] (fused step only)
    MgGenOp* s_ops = reinterpret_cast<MgGenOp*>(s_shared + lc.sh.ops);                  // [kOpsLds] the reset program's first ops (fused step)
    // (mg_step_render_encode's instantiations) [fs.enc_ne] grid byte -> (type, colour, state): behind the last wave's scratch
    uint32_t* s_enc = reinterpret_cast<uint32_t*>(s_shared + lc.sh.total + (size_t)WPB * lc.L.total);
    constexpr bool kChunkRaster = TS_ > 0 && (TS_ % 8) == 0 && RM_ == 0;
    constexpr bool kStreamRaster = !kChunkRaster && !kGather;     // assemble-and-stream
    // the rasters bound by instruction issue run at a raised wave priority (phase 6); measured per instantiation: the gather
    // raster at 11-pixel tiles without 'prestige' agents is close enough to the HBM bound to lose 0.6 % by it
    constexpr bool kRasterPrio = (kGather && (TS_ <= 10 || V_ == 9)) || kStreamRaster;
    // The wave's scratch pointers (w_stage_g ... w_out), the launch's dimensions (VS, TS, n, nv: the viewers —
    // all n agents by default; a subset when the env's agents differ in view size / tile size / offset and are
    // rendered group by group, agents.py:19-35) and the dividers have ONE definition: MG_REGION_LOCALS, expanded
    // in every region that uses them (lc.L = render_scratch_for(cfg, WPB, RM_), worked out by the launcher).

    // raster geometry of the 16-byte-chunk path (see phase 6): pairs per pixel row, a lane's start
    // position and its per-trip advance — constants of the launch, folded at compile time when VS_ > 0
    const uint32_t PR = (uint32_t)(VS_ ? VS_ : cfg.view_size) * (uint32_t)((TS_ ? TS_ : cfg.tile_size) * 3 / 8);
    constexpr uint32_t CH_STRIDE = kWave;                             // chunks a wave advances per trip
    const uint32_t c_first = (uint32_t)lane;
    const uint32_t STEP_R = PR ? (2u * CH_STRIDE) / PR : 0u, STEP_P = PR ? (2u * CH_STRIDE) - STEP_R * PR : 0u;
    const uint32_t rast_r0 = PR ? (2u * c_first) / PR : 0u, rast_p0 = PR ? 2u * c_first - rast_r0 * PR : 0u;

    // The inputs of a wave's run — 240 B of grid and a few agent records per env — are staged in LDS a BATCH of
    // envs at a time: on gfx950 the wait for a load's data is a vmcnt wait, vmcnt is in-order and also counts this
    // wave's stores, so every load consumed in the middle of the run drains the wave's whole store queue first
    // (the pure-store microbenchmark loses 25 % of its throughput to one such load per env).  At the bench batch
    // a wave's run is one batch: it reads before its first store and never again.
    const int K = lc.L.stage_envs;
    MG_STAMP(13);

    // assemble-and-stream raster: the wave's run of envs is ONE contiguous output stream.  w_out[0] is
    // the byte at the 16-byte-aligned global address out_base; w_out[0 .. carry) are pending bytes of a
    // chunk that is not complete yet (at the start of the run: `head` bytes that belong to the wave before)
    // (a POINTER derived from `obs`, not an integer: the compiler then knows the address space and emits
    // global_store — through uintptr_t it emitted flat_store, which also counts in lgkmcnt, the LDS counter)
    uint8_t* out_base = obs;
    uint32_t carry = 0, head = 0;
    // How many envs' views a wave derives together before it rasters them (see the env loop).  Chunk raster (HBM-bound;
    // a wave's stores stop while it is in the view phases, so the waves of a workgroup should not all be there at once, and
    // a group's one shadow cast serves all its envs): groups of TWO at tile 8, env by env at tile 16 / 32 — measured against
    // one, three, four and against round 3's scheme (1 / 2 / 4 / 8 by wave behind a ramp 1, 2, 4): the whole step -0.96 %
    // (`profiles/r04/ab_fused_depth_policies_v33.txt`, `ab_render_depth_*`); the 'prestige' variants keep that scheme.  The
    // assemble-and-stream rasters are not HBM-bound: the views of the whole staged batch at once (tile 11: +1.5 % against
    // the by-wave depths; the 'prestige' variants are bound by the latency of the view phases — one or three viewers leave
    // most lanes of a trip idle when the envs are taken one at a time; their per-env recoloured tiles, which have ONE slot,
    // are made right before the env's raster, phase 4b).  depth_mode > 0 (measurement builds) forces one depth for all.
    int depth;
    {
        MG_REGION_LOCALS;
        if constexpr (kGather && kPrestige) {       // the recoloured tiles' rows are padded like the atlas's: the zeros, once
            for (int i = lane; i < (L.out - L.dyn) / 16; i += kWave) reinterpret_cast<uint4*>(w_dyn)[i] = make_uint4(0, 0, 0, 0);
            wave_lds_sync();
        }
        if constexpr (kStreamRaster) {
            const size_t a0 = (size_t)e0 * nv * img_bytes;
            head = (uint32_t)((reinterpret_cast<uintptr_t>(obs) + a0) & 15);
            carry = head;
            out_base = obs + a0 - head;
            for (int i = lane; i < L.out_chunks; i += kWave) reinterpret_cast<uint4*>(w_out)[i] = make_uint4(0, 0, 0, 0);
            wave_lds_sync();
        }
        // (the gather raster: groups of FOUR — measured against 1, 2, 3 and the whole batch of 8, raster alone: -4.0 % at tile 5,
        // -3.8 % at tile 6 against 8; with 8 the first store of a launch waits for 23 us of views, profiles/r05)
        // (round 6, with the views in row form — a lane per view row —: at 5-pixel tiles groups of TWO for views 7 and 9, whose four
        // envs would be 84 / 108 rows — a full trip and a fifth of one —: raster alone 0.0727 -> 0.0711 ms at view 7, 0.0610 ->
        // 0.0582 at view 9; every other gather geometry stays at four: `profiles/r06/ab_render_view_group_depth_sweep_v9.txt`)
        constexpr int kGatherDepth = (TS_ == 5 && (VS_ == 7 || VS_ == 9) && !kPrestige) ? 2 : 4;
        depth = depth_mode > 0 ? depth_mode : kGather ? kGatherDepth : !kChunkRaster ? L.tmap_slots : kPrestige ? (1 << (wave & 3)) : (TS_ == 8 ? 2 : 1);
        if (depth > L.tmap_slots) depth = L.tmap_slots;
        if (depth > L.view_slots) depth = L.view_slots;   // (a group's views need a scratch slot per env)
    }
    // item -> (slot, rest), view cell -> (viewer, row, column): 24-bit multiplies only (Div20; MG_REGION_LOCALS)
    constexpr bool kExactVV = VS_ > 0 && VS_ <= 9;     // 16 viewers * VS^2 cells: x * (m*d - 2^20) < 2^20 holds (checked below)
    static_assert(VS_ == 0 || VS_ > 9 || (16u * VS_ * VS_ * ((((1u << 20) + VS_ * VS_ - 1u) / (VS_ * VS_)) * (VS_ * VS_) - (1u << 20)) < (1u << 20)), "Div20 exactness");

    for (int eb = e0; ; eb += K) {
        const bool first = (eb == e0);
        if (eb >= e_end && !first) break;
        const int kb = max(0, min(K, e_end - eb));
        MG_REGION_LOCALS;
        // 0. stage the batch (contiguous in HBM): loads first, all in flight, then one wait.  mg_step_render:
        //    the step's own up-front loads ride the same round trip, the envs are stepped on the staged grids
        //    (lane j: env eb + j) and their records are staged from the step's scratch.
        //    The wave's FIRST pass is also the launch's PROLOGUE: tables + atlas -> LDS ride the same round trip,
        //    and the workgroup's one barrier follows — every wave of the workgroup passes here exactly once, with
        //    or without envs of its own.  (tools/phase_stamps.py: with the atlas copied first and the batch staged
        //    after the barrier, the median wave waited 2 us + 5.5 us — 8.8 us with the env step's loads — before
        //    it could begin.)  ONE round trip has to be defended against the compiler: nothing may be computed on
        //    a loaded value before the last request is out (a select, a sign extension, the merge of two widths
        //    into one variable: each was a vmcnt(0) in the middle — round 2's prologue was five round trips).
        //    Tried on top of this and measured in the product build (profiles/r03/ab_head_variants_v18.txt):
        //    the inputs TOUCHED ahead (two LDS-DMA loads per wave, before the launch's constants are set up, so that
        //    this round trip ends in the L2): +0.7 %; the operands of the RNG head's top-up (mt_finish) requested
        //    here and handed to the step through LDS: +2 % — what they save at the end of step_run they cost in
        //    front of it; both removed.
        //    (A workgroup-wide variant — the 128 envs of a batch stepped by two full waves between two
        //    barriers instead of 8 lanes in each of 16 waves — was measured 4 us SLOWER per launch: the step is
        //    bound by the latency of one wave's dependent chain, not by issue slots, and a 64-lane wave runs
        //    the union of its lanes' branches — nearly always including a reset.)
        //    (LDS-DMA for the grids, the atlas and the object table — no staging registers — was tried here: hipcc
        //    then waits vmcnt(0) at every use of an ordinary load's result and before the barrier, which turns the
        //    one round trip into eight.)
        {
            // (`tidl`, `lanel`: the thread and lane index, opaque to the compiler — everything derived from them in here
            // is an invariant of the env loop to it, computed ahead of the loop, spilled at the 128-VGPR limit and
            // reloaded with a full wait in the middle of this round trip)
            int tidl = tid, lanel = lane;
            asm volatile("" : "+v"(tidl), "+v"(lanel));
            if (first) MG_STAMP(14);
            const int T = WPB * 64;
            const uint32_t* gsrc = reinterpret_cast<const uint32_t*>(st.grid + (size_t)eb * cfg.cells_stride);
            const uint64_t* rsrc = st.agents + (size_t)eb * n;
            const int nd = kBigGrid ? 0 : kb * gdw, nr = kb * n;
            // (16 bytes per lane and request: a batch's grids are whole 16-byte chunks — cells_stride is a multiple of 16 —,
            // and every request costs its address, its bounds check and its exec mask: 2 instead of 8)
            static_assert(kSR % 4 == 0, "grid dwords per lane: whole uint4s");
            uint4 v[kSR / 4];
#pragma unroll
            for (int q = 0; q < kSR / 4; q++) {
                const int i = q * kWave + lanel;
                v[q] = 4 * i < nd ? reinterpret_cast<const uint4*>(gsrc)[i] : make_uint4(0, 0, 0, 0);
            }
            const int r0i = lanel, r1i = lanel + kWave;                     // kb * n <= 8 * 16 = 2 * kWave records
            uint64_t rv0 = 0ull, rv1 = 0ull;
            double pv0 = 0., pv1 = 0.;
            if (!fs.enabled) {
                rv0 = r0i < nr ? rsrc[r0i] : 0ull; rv1 = r1i < nr ? rsrc[r1i] : 0ull;
                if constexpr (kPrestige) {
                    const double* psrc = st.prestige + (size_t)eb * n;
                    if (r0i < nr) pv0 = psrc[r0i];
                    if (r1i < nr) pv1 = psrc[r1i];
                }
            }
            const uint4* asrc = reinterpret_cast<const uint4*>(cfg.atlas);
            const int na = kGlobalAtlas ? 0 : render_atlas_raw_bytes(cfg) / 16;   // (0 when the atlas is read in place)
            const int no = fs.enabled ? cfg.n_obj * 2 : 0;                  // object table (fused step): <= 512 16-byte pieces
            const int nops = fs.enabled && fs.has_prog ? min(fs.prog.n_ops, kOpsLds) * 2 : 0;   // ... and the reset program's first ops
            uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, o0 = a0, o1 = a0, p0 = a0;
            uint8_t f = 0, sl = 0xFF, f2 = 0, vmap0 = 0;
            uint32_t hide0 = 0, enc_raw = 0, enc_col = 0;
            double pscale0 = 0.;
            // padded tile rows (kPadRows): LDS dword d is 4 bytes of row d / kRowW, read as the 8 aligned bytes of the
            // atlas around them (pad_source, mg_gather.h) and cut out when they are stored
            const int raw16 = render_atlas_raw_bytes(cfg), pad_rows = 4 * cfg.n_tiles * TS, npd = kPadRows ? pad_rows * kRowW + kPadTailW : 0;
            (void)raw16; (void)pad_rows;
            uint2 pw[kPadQ];
#pragma unroll
            for (int q = 0; q < kPadQ; q++) pw[q] = make_uint2(0, 0);
            // (the host's copy of the atlas in this layout, when there is one — MgConfig::
            // atlas_gather_off —, as plain 16-byte pieces: three per thread in this round trip, in the same registers)
            const int goff = kPadRows ? cfg.atlas_gather_off : 0;
            const int npq = (npd + 3) / 4;
            static_assert(kPadQ == 6, "three uint4 of the ready-made atlas ride in pw[0..5]");
            if (first) {
                if constexpr (kPadRows) {
                    if (goff) {
                        const uint4* psrc = reinterpret_cast<const uint4*>(cfg.atlas + goff);
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            if (tidl + j * T < npq) {
                                const uint4 t4 = psrc[tidl + j * T];
                                pw[2 * j] = make_uint2(t4.x, t4.y);
                                pw[2 * j + 1] = make_uint2(t4.z, t4.w);
                            }
                    } else {
#pragma unroll
                    for (int q = 0; q < kPadQ; q++) {
                        uint32_t cut, keep;
                        const int a = tidl + q * T < npd ? pad_source<3 * TS_, kPadFrontW, kRowW>(tidl + q * T, pad_rows, raw16, cut, keep) : -1;
                        if (a >= 0) pw[q] = *reinterpret_cast<const uint2*>(cfg.atlas + a);
                    }
                    }
                } else {
                    if (tidl < na) a0 = asrc[tidl];
                    if (tidl + T < na) a1 = asrc[tidl + T];
                }
                if (tidl < no) o0 = reinterpret_cast<const uint4*>(cfg.obj)[tidl];
                if (tidl + T < no) o1 = reinterpret_cast<const uint4*>(cfg.obj)[tidl + T];      // (more than 128 kinds at 4 waves)
                if (tidl < nops) p0 = reinterpret_cast<const uint4*>(fs.prog.ops)[tidl];
                if (tidl < cfg.n_obj) {         // (T >= 256 >= n_obj: one trip)
                    f = cfg.obj[tidl].flags; sl = cfg.obj[tidl].ovl_slot; f2 = cfg.obj[tidl].flags2;
                    if (cfg.any_hide && cfg.hide_by_obj) hide0 = cfg.hide_by_obj[tidl];
                }
                if (tidl < MG_MAX_AGENTS) {     // the per-agent tables of the launch struct, requested with the rest
                    pscale0 = cfg.prestige_scale[tidl];
                    vmap0 = cfg.view_agent[tidl];
                }
                if constexpr (kEnc) {
                    if (fs.encode_out && tidl < fs.enc_ne) {     // (mg_step_render_encode) the encode table's sources: T >= 256 >= enc_ne
                        if (tidl < cfg.n_obj) enc_raw = *reinterpret_cast<const uint32_t*>(cfg.obj + tidl);
                        else if (tidl - cfg.n_obj < 4 * cfg.n_agents) enc_col = cfg.agent_color_idx[(tidl - cfg.n_obj) >> 2];
                    }
                }
            }
            StepScratch sc;
            if (fs.enabled) sc = fused_step_scratch(cfg, lane, ws + L.step, s_obj, s_oflags);
            StepEnv se = {0, 0};
#if defined(MG_AB_VARIANTS)
            sc.stamp = (d_ab_stamps && lane == 0) ? d_ab_stamps + (size_t)(blockIdx.x * WPB + wave) * 24 + (first ? 8 : 16) : nullptr;
#endif
            if (fs.enabled) {
                const MgState& st = kernarg_again<MgState>(offsetof(RenderKernargs, st));
                const FusedStep& fs = kernarg_again<FusedStep>(offsetof(RenderKernargs, fs));
                const StepLoadRegs r = step_load_issue(cfg, st, fs.actions, fs.action_bytes, eb, kb, lanel);
                __builtin_amdgcn_sched_barrier(0);
                se = step_load_commit(cfg, r, kb, lanel, sc, by_n);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (first) MG_STAMP(15);
#pragma unroll
            for (int q = 0; q < kSR / 4; q++) {
                const int i = q * kWave + lanel;
                if (4 * i < nd) reinterpret_cast<uint4*>(w_stage_g)[i] = v[q];
            }
            for (int i = kSR * kWave + lanel; i < nd; i += kWave) reinterpret_cast<uint32_t*>(w_stage_g)[i] = gsrc[i];   // (grids beyond 2 KiB per batch: a second trip)
            if (!fs.enabled) {
                if (r0i < nr) { const int j = (int)by_n.div((uint32_t)r0i); w_stage_r[j * rec_stride + (r0i - j * n)] = rv0; if constexpr (kPrestige) w_stage_p[j * rec_stride + (r0i - j * n)] = pv0; }
                if (r1i < nr) { const int j = (int)by_n.div((uint32_t)r1i); w_stage_r[j * rec_stride + (r1i - j * n)] = rv1; if constexpr (kPrestige) w_stage_p[j * rec_stride + (r1i - j * n)] = pv1; }
            }
            if (first) {
                if constexpr (kPadRows) {
                    if (goff) {
                        uint4* adst4 = reinterpret_cast<uint4*>(s_atlas);
                        const uint4* psrc = reinterpret_cast<const uint4*>(cfg.atlas + goff);
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            if (tidl + j * T < npq) adst4[tidl + j * T] = make_uint4(pw[2 * j].x, pw[2 * j].y, pw[2 * j + 1].x, pw[2 * j + 1].y);
                        for (int i0 = tidl + 3 * T; i0 < npq; i0 += 4 * T) {
                            uint4 w[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) w[u] = i0 + u * T < npq ? psrc[i0 + u * T] : make_uint4(0, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (i0 + u * T < npq) adst4[i0 + u * T] = w[u];
                        }
                    } else {
                    uint32_t* adst = reinterpret_cast<uint32_t*>(s_atlas);
#pragma unroll
                    for (int q = 0; q < kPadQ; q++) {
                        const int d = tidl + q * T;
                        uint32_t cut = 0, keep = 0;
                        if (d < npd) adst[d] = pad_source<3 * TS_, kPadFrontW, kRowW>(d, pad_rows, raw16, cut, keep) >= 0 ? pad_cut(pw[q].x, pw[q].y, cut, keep) : 0u;
                    }
                    // (larger atlases: the rest, eight requests per thread and round trip — as one load and one store per iteration
                    // the 45 dwords per thread of the reference's example, 11-pixel tiles with 'prestige' sprites, were 39 dependent
                    // round trips: 13 us of a 134 us launch before its workgroups' barrier)
                    constexpr int kPadU = 8;         // (12 or 16 per trip spill registers even in the 12-wave variants)
                    for (int d0 = tidl + kPadQ * T; d0 < npd; d0 += kPadU * T) {
                        uint2 w[kPadU];
                        uint32_t cut[kPadU], keep[kPadU];
                        int a[kPadU];
#pragma unroll
                        for (int u = 0; u < kPadU; u++) {
                            const int d = d0 + u * T;
                            cut[u] = keep[u] = 0;
                            a[u] = d < npd ? pad_source<3 * TS_, kPadFrontW, kRowW>(d, pad_rows, raw16, cut[u], keep[u]) : -1;
                            w[u] = make_uint2(0, 0);
                            if (a[u] >= 0) w[u] = *reinterpret_cast<const uint2*>(cfg.atlas + a[u]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < kPadU; u++) {
                            const int d = d0 + u * T;
                            if (d < npd) adst[d] = a[u] >= 0 ? pad_cut(w[u].x, w[u].y, cut[u], keep[u]) : 0u;
                        }
                    }
                    }
                } else {
                uint4* adst = reinterpret_cast<uint4*>(s_atlas);
                if (tidl < na) adst[tidl] = a0;
                if (tidl + T < na) adst[tidl + T] = a1;
                if constexpr (TS_ == 8) {      // (8-pixel tiles: 192 bytes per tile — the atlas of 28 tiles is 21 KB: no such rest unless there are 40 kinds)
                    for (int i = tidl + 2 * T; i < na; i += T) adst[i] = asrc[i];
                } else
                for (int i0 = tidl + 2 * T; i0 < na; i0 += 4 * T) {             // (larger atlases: the rest, four requests per thread and round trip)
                    uint4 w[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) w[u] = i0 + u * T < na ? asrc[i0 + u * T] : make_uint4(0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (i0 + u * T < na) adst[i0 + u * T] = w[u];
                }
                }
                if (tidl < no) reinterpret_cast<uint4*>(s_obj)[tidl] = o0;
                if (tidl + T < no) reinterpret_cast<uint4*>(s_obj)[tidl + T] = o1;
                if (tidl < nops) reinterpret_cast<uint4*>(s_ops)[tidl] = p0;
                // (anything computed on a loaded value goes here, behind the round trip: in front of the step's loads
                // it was a wait for the first half of the requests before the second half was issued)
                if (tidl == 0) { f = MG_OF_SEE_BEHIND | MG_OF_CAN_OVERLAP; sl = 0; f2 = 1; }   // empty cell
                if (tidl < NO) {
                    s_oflags[tidl] = f;
                    s_oslot[tidl] = sl;
                    s_oflags2[tidl] = f2;
                    if (cfg.any_hide) s_hideby[tidl] = hide0;
                }
                if (tidl < MG_MAX_AGENTS) {
                    s_pscale[tidl] = pscale0;
                    s_vmap[tidl] = cfg.n_view ? vmap0 : (uint8_t)tidl;
                }
                if constexpr (kEnc) {
                    if (fs.encode_out && tidl < fs.enc_ne) {
                        // id 0 (None) and the padding encode as zeros; agent code 4 k + d: (agent_type_idx, colour of k, d)
                        const int code = tidl - cfg.n_obj;
                        s_enc[tidl] = tidl < cfg.n_obj ? (tidl ? enc_raw & 0xFFFFFFu : 0u)
                                    : code < 4 * cfg.n_agents ? ((uint32_t)cfg.agent_type_idx | (enc_col << 8) | ((uint32_t)(code & 3) << 16)) : 0u;
                    }
                }
            }
            if (first) {
                __syncthreads();
                MG_STAMP(1);
            }
            if (eb >= e_end) break;
            if (eb == e0) MG_STAMP(2);
            if (fs.enabled) {
                MG_REGION_LOCALS;
                const MgState& st = kernarg_again<MgState>(offsetof(RenderKernargs, st));
                const FusedStep& fs = kernarg_again<FusedStep>(offsetof(RenderKernargs, fs));
                // (The whole batch is stepped here, before its first store.  Stepping group by group — each view
                // group right before its views, so that the wave's first store waits for the step of ONE env — was
                // measured 11 % SLOWER (profiles/r03/ab_fused_step_per_group*.txt): the step's RNG refill is a
                // dependent HBM load, and a load issued after stores waits for the wave's whole store queue.)
                wave_lds_sync();
                bool wrote = false;
                int head_k = 0;
                sc.defer_writeback = true;      // records and RNG heads go back to HBM from the whole wave, below
                // one-round-trip head refills through LDS-DMA (mt_generate16_dma): their landing zone, 9 rows of 128 bytes,
                // is the tmap area — the views have not begun
                sc.dma = (size_t)L.tmap_slots * L.tmap_stride >= kMtDmaBufDwords * 4 ? reinterpret_cast<uint32_t*>(w_tmap0) : nullptr;
                // The step in three parts (mg_core.h): step_begin on the env's lane (late spawns, the shuffle); the agents
                // resolved by ONE LANE PER (agent, env) on the pre-step state — lane 8 k + j: agent k of staged env j, up
                // to 8 agents — instead of a sequential loop of ~1 700 dependent instructions on the env's lane, which then
                // only runs for an env whose agents' actions do depend on their order (step_par_commit says which);
                // step_end on the env's lane (done agents, respawn, the episode's end and the reset that follows it).
                StepCtx ctx;
                // (the staged grids of the batch — or, for a grid that is read in place, the batch's grids where they live)
                uint8_t* const g_batch = kBigGrid ? st.grid + (size_t)eb * cfg.cells_stride : w_stage_g;
                uint8_t* const g_mine = g_batch + (size_t)(lane & 7) * cfg.cells_stride;
                if (lane < kb) ctx = step_begin(cfg, st, eb + lane, se, sc, g_mine);
                bool loop = true;                       // this env's agents take the sequential loop
                if (n <= 8) {
                    if (lane < kb) step_par_publish(cfg, sc, ctx);
                    wave_lds_sync();
                    // (`lp`: the lane index, opaque to the compiler — what step_par_* derive from it, agent = lane / 8 and env =
                    // lane % 8, is otherwise an invariant of the env loop: computed at the kernel's entry, spilled at the
                    // register limit and fetched back from scratch memory four times per batch)
                    int lp = lane;
                    asm volatile("" : "+v"(lp));
                    const ParLane P = step_par_resolve(cfg, sc, g_batch, kb, lp);
                    wave_lds_sync();
                    loop = step_par_commit(cfg, st, fs.rewards, eb, sc, P, lp);
                    wave_lds_sync();
#if defined(MG_AB_VARIANTS)
                    if (sc.stamp && !loop) sc.stamp[2] = wall_clock64();      // (the stamp step_agents sets when it runs)
#endif
                }
                if (lane < kb) {
                    if (loop) step_agents(cfg, st, fs.rewards, eb + lane, sc, g_mine, ctx);
                    // (the reset program's ops from their LDS copy when all of them are there — the prologue staged the first
                    // kOpsLds —, else in place)
                    MgGenProgram prog = fs.prog;
                    if (prog.n_ops <= kOpsLds) prog.ops = s_ops;
                    const StepOut so = step_end(cfg, st, prog, fs.has_prog != 0, eb + lane, sc, g_mine, ctx);
                    wrote = so.wrote;
                    head_k = so.head_k;
                    if constexpr (kPrestige) {
                        // agent.prestige as this lane left it in HBM, and — here, in the latency-bound step part where
                        // the VALU is idle, one lane per ENV — the colour it gives the agent's sprite (the float64
                        // tanh of render_post, agents.py:92-119): the raster's per-env phase only reads it
                        for (int k = 0; k < n; k++) {
                            const double p = st.prestige[(size_t)(eb + lane) * n + k];
                            w_stage_p[lane * rec_stride + k] = p;
                            if ((cfg.prestige_mask >> k) & 1u) {
                                const PrestigeColor c = prestige_color(p, s_pscale[k]);
                                w_stage_c[lane * rec_stride + k] = c.r | (c.g << 8) | (c.b << 16);
                            }
                        }
                    }
                }
                uint64_t todo = __ballot(wrote);
                wave_lds_sync();
                // The stepped records (to HBM, and to where the views read them) and the RNG heads (a ring per env that
                // starts at head_k: mt_finish_ring), by the whole wave as the contiguous runs they are: two coalesced
                // stores per lane instead of sixteen scattered ones — and as many LDS reads — in every stepping lane.
                {
                    uint64_t* rdst = st.agents + (size_t)eb * n;
                    uint32_t* hdst = st.mt_head + (size_t)eb * MG_MT_HEAD;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int i = lane + q * kWave;
                        const int e = i >> 4, j = i & (MG_MT_HEAD - 1);             // head word j of env e
                        const int ke = __shfl(head_k, e & 7);
                        if (i < kb * MG_MT_HEAD) hdst[i] = sc.head[((j + ke) & (MG_MT_HEAD - 1)) * 8 + e];
                        if (i < kb * n) {
                            const int er = (int)by_n.div((uint32_t)i), k = i - __mul24(er, n);
                            const uint64_t r = sc.rec[k * 8 + er];
                            rdst[i] = r;
                            w_stage_r[er * rec_stride + k] = r;
                        }
                    }
                }
                if constexpr (kBigGrid) todo = 0;       // (written where they live)
                while (todo) {      // grid slices the step wrote: back to HBM, the whole wave per slice
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(w_stage_g + (size_t)j * cfg.cells_stride);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(st.grid + (size_t)(eb + j) * cfg.cells_stride);
                    for (int i = lane; i < gdw; i += kWave) dst[i] = src[i];
                }
            }
        }
        wave_lds_sync();
        if (eb == e0) MG_STAMP(3);
    // `depth` envs at a time: first all their views (phases 1-5 -> one tmap slot each), then all their rasters (how many:
    // where `depth` is chosen, above).  The 'prestige' chunk-raster variants RAMP in their first batch — the first env
    // alone, so that the wave's first store leaves early, then groups of 2, 4, ... up to the wave's depth.
    const bool ramp = kChunkRaster && kPrestige && depth_mode <= 0 && eb == e0;
    int gd = ramp ? 1 : depth;                  // size of the current group
    // mg_step_render_encode: MultiGrid.encode (base.py:196-214) of the batch this wave has just stepped — its stepped grids and
    // records are staged: the first agent of an empty cell marks it in the grid bytes, then a lane per aligned 16-byte chunk of
    // the batch's part of the flat output stream (mg_encode_core.h); +2.4 % bytes on the bench's shape.  WHEN: behind the
    // rasters of the batch's first group — the wave's first image stores are out, the other waves' stores cover the look-ups'
    // latency (at the end of the wave's run it was the launch's tail: +4.1 % instead of the bytes' +2.4 %), and the marks are
    // taken out again for the views that follow; a batch of one group: when it is done.
    bool enc_pending = false;
    if constexpr (kEnc) enc_pending = fs.enabled && fs.encode_out != nullptr && kb > 0;
    auto encode_batch = [&](const bool undo) {
        if constexpr (!kEnc) { (void)undo; return; } else {
        MG_REGION_LOCALS;
        const FusedStep& fse = kernarg_again<FusedStep>(offsetof(RenderKernargs, fs));
        int le = lane;
        asm volatile("" : "+v"(le));
        encode_batch_mark(cfg, w_stage_g, w_stage_r, rec_stride, kb, fse.enc_m_n, le, kWave);
        wave_lds_sync();
        encode_batch_chunks(cfg, w_stage_g, s_enc, fse.encode_out, (long long)eb * (W * H), kb, W * H, fse.enc_m_cells, le, kWave);
        wave_lds_sync();
        if (undo) {
            encode_batch_mark(cfg, w_stage_g, w_stage_r, rec_stride, kb, fse.enc_m_n, le, kWave, true);
            wave_lds_sync();
        }
        enc_pending = false;
        }
    };
    (void)encode_batch; (void)enc_pending;
    for (int ej0 = 0; ej0 < kb; ej0 += gd, gd = ramp ? min(2 * gd, L.view_slots) : depth)
    for (int pass = 0; pass < 2; pass++)
    for (int ej = ej0; ej < min(kb, ej0 + gd); ej++) {
        if constexpr (kEnc) { if (enc_pending && ej0 > 0) encode_batch(true); }
        MG_REGION_LOCALS;
        const int e = eb + ej;
        uint16_t* w_tmap = w_tmap0 + (size_t)(ej - ej0) * (L.tmap_stride / 2);
        if (pass == 0) {
        // Views of the group's G envs at once (phases 1-5 -> one tmap slot each; G: where `depth` is chosen).  Phases 2
        // and 4 have one lane per AGENT / VIEWER — a lone env leaves 61 of 64 lanes idle — and 3 x 49 view cells fill
        // 2.3 trips of 64 lanes where 8 envs fill 18.4 of 19.
        if (ej != ej0) continue;
        const int G = min(kb, ej0 + gd) - ej0;
        const int nvVV = nv * VV;
        const uint8_t* g_grid = kBigGrid ? kernarg_again<MgState>(offsetof(RenderKernargs, st)).grid + (size_t)e * cfg.cells_stride
                                         : w_stage_g + (size_t)ej * cfg.cells_stride;      // env ej + g: + g * cells_stride
        const uint64_t* g_rec = w_stage_r + (size_t)ej * rec_stride;            //            + g * rec_stride
        // 1. scratch of the G slots
        const bool has_second = cfg.any_hide;                    // (no `second` slots without hide_item_types)
        if constexpr (!kBigGrid)
            for (int i = lane; i < G * (L.cell_stride / 4); i += kWave) {
                reinterpret_cast<uint32_t*>(w_first)[i] = 0xFFFFFFFFu;
                if (has_second) reinterpret_cast<uint32_t*>(w_second)[i] = 0xFFFFFFFFu;
            }
        for (int i = lane; i < G * L.trow_stride; i += kWave) w_trow[i] = 0;
        wave_lds_sync();
        if constexpr (V_ == 3 || V_ == 4) {
            for (int g = 0; g < G; g++)
                for (int it = lane; it < nv * VV; it += kWave) w_tmap[g * (L.tmap_stride / 2) + it] = 0;
        } else {
        for (int rep = 0; rep < (V_ == 11 ? 2 : 1); rep++) {   // V_ == 11 (measurement): phases 2-5 twice
        // 2. first (lowest-rank) agent of every occupied cell: the reference's "cell object" when
        //    the base is empty, and `obj.agents[0]` when agents stand on an overlappable object
        auto first_of_cell = [&](const int g, const int a) {     // slot g, agent a
            const uint64_t* w_rec = g_rec + g * rec_stride;
            const uint64_t r = w_rec[a];
            if (rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED) {
                int below = 0;   // agents of this cell that arrived earlier
                for (int j = 0; j < n; j++) {
                    const uint64_t rj = w_rec[j];
                    if (j != a && (rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == rec_xy(r) &&
                        rec_byte(rj, MG_AG_RANK) < rec_byte(r, MG_AG_RANK))
                        below++;
                }
                const int cell = g * L.cell_stride + rec_byte(r, MG_AG_X) * H + rec_byte(r, MG_AG_Y);
                if (below == 0) w_first[cell] = (uint8_t)a;
                else if (below == 1 && has_second) w_second[cell] = (uint8_t)a;   // only hide_item_types looks at it
            }
        };
        if constexpr (!kBigGrid)
            for (int it = lane; it < G * n; it += kWave) { const int g = (int)by_n.div((uint32_t)it); first_of_cell(g, it - __mul24(g, n)); }
        // 2b. one lane per VIEWER: its view as an affine map of (column va, row vb) — SURVEY.md A.4's four cases
        //     folded into an origin, a swap bit and two signs — and who it is, so that phase 3 does no per-cell
        //     case analysis (as nested branches it ran every lane through all four headings):
        //       p = swap ? vb : va, q = swap ? va : vb;  wx = x0 +- p;  wy = y0 +- q
        //     word 0: x0 + 256 | (y0 + 256) << 10 | swap << 20 | negx << 21 | negy << 22
        //     word 1: x | y << 8 | agent << 16 | orientation (3 - dir) & 3 << 24 | in the grid << 26
        auto view_affine = [&](const int g, const int v) {
            const uint32_t k = s_vmap[v];
            const uint64_t r = g_rec[__mul24(g, rec_stride) + (int)k];
            const int x = (int)rec_byte(r, MG_AG_X), y = (int)rec_byte(r, MG_AG_Y), dir = (int)rec_byte(r, MG_AG_DIR);
            int x0, y0;
            uint32_t bits;
            if (dir == 3)      { x0 = x - h;                 y0 = y - (VS - 1) + off;  bits = 0u; }
            else if (dir == 0) { x0 = x - off + (VS - 1);    y0 = y - h;               bits = 1u | 2u; }
            else if (dir == 1) { x0 = x - h + (VS - 1);      y0 = y - off + (VS - 1);  bits = 2u | 4u; }
            else               { x0 = x - VS + 1 + off;      y0 = y - h + (VS - 1);    bits = 1u | 4u; }
            const uint32_t placed = (rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED) ? 1u : 0u;   // (an evicted viewer is in no stack)
            w_vaff[__mul24(g, nv) + v] = make_uint2((uint32_t)(x0 + 256) | ((uint32_t)(y0 + 256) << 10) | (bits << 20),
                                                    (uint32_t)x | ((uint32_t)y << 8) | (k << 16) | (((3u - (uint32_t)dir) & 3u) << 24) | (placed << 26));
        };
        for (int it = lane; it < G * nv; it += kWave) { const int g = (int)by_nv.div((uint32_t)it); view_affine(g, it - __mul24(g, nv)); }
        wave_lds_sync();
        // 3. egocentric crop + rotate: view cell (a = column, b = row) -> world cell; the cell's object, the
        //    agent shown on it, its transparency, and — tile selection (base.py:275-299) as if the cell were visible — the atlas
        //    offset of the view cell (phase 5 puts the shadow tile where it is not).  All index arithmetic in 24-bit
        //    multiplies (Div20; offsets of slot g are products of small numbers).
        //    view_cell: ONE cell, everything about its slot and viewer handed in.  Written without branches around its
        //    look-ups (an out-of-grid cell reads cell 0 and discards it): the row form below unrolls it over a view row, and the
        //    look-ups of the row's cells then travel together instead of one dependent round trip after the other.
        auto view_cell = [&](const uint32_t g, const uint32_t v, const uint32_t va, const uint32_t vb, const int wx, const int wy,
                             const uint8_t* w_grid, const uint32_t gcell, const uint2 aff, const uint8_t* w_recb, uint16_t* tmap) -> bool {
            const uint32_t k = (aff.y >> 16) & 0xFFu;
            const int x = (int)(aff.y & 0xFFu), y = (int)((aff.y >> 8) & 0xFFu);
            const bool inb = (uint32_t)wx < (uint32_t)W && (uint32_t)wy < (uint32_t)H;
            const int cell = inb ? __mul24(wx, H) + wy : 0;
            uint32_t base = w_grid[cell], show;
            uint32_t big1 = 0xFF, big2 = 0xFF;          // kBigGrid: the first / second agent of the cell, searched
            if constexpr (kBigGrid) {
                // the lowest- and second-lowest-rank agents standing on the cell, among the env's n (what the per-cell maps
                // `first` / `second` hold for grids that fit LDS)
                const uint64_t* w_rec = g_rec + __umul24(g, (uint32_t)rec_stride);
                const uint32_t cxy = (uint32_t)wx | ((uint32_t)wy << 8);
                uint32_t r1 = 0xFF, r2 = 0xFF;
                for (int j = 0; j < n; j++) {
                    const uint64_t rj = w_rec[j];
                    if (inb && (rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == cxy) {
                        const uint32_t rk = rec_byte(rj, MG_AG_RANK);
                        if (rk < r1) { r2 = r1; big2 = big1; r1 = rk; big1 = (uint32_t)j; }
                        else if (rk < r2) { r2 = rk; big2 = (uint32_t)j; }
                    }
                }
                show = big1;
            } else show = w_first[gcell + cell];
            if (!inb) { base = 0; show = 0xFF; }
            if (show != 0xFF && wx == x && wy == y && ((aff.y >> 26) & 1u)) show = k;   // viewer in the stack: base.py:282-291
            const bool see = (s_oflags[base] & MG_OF_SEE_BEHIND) != 0;                  // opacity first
            if (cfg.any_hide && inb) {
                // hide_item_types (base.py:441-449), applied after visibility: a hidden cell object is
                // replaced by the first agent standing on it (or nothing) and that agent is drawn as
                // a plain cell object — the "viewer is in the stack" rule no longer applies to it
                const uint32_t first = kBigGrid ? big1 : (uint32_t)w_first[gcell + cell];
                if (base && ((s_hideby[base] >> k) & 1u)) { base = 0; show = first; }
                else if (base == 0 && first != 0xFF && first != k && ((cfg.hide_agent_mask >> k) & 1u))
                    show = kBigGrid ? big2 : (uint32_t)w_second[gcell + cell];
            }
            // (the tile is selected HERE, where the cell's object, agent and viewer are in registers: as a per-cell phase of its
            // own behind the shadow cast it re-derived the cell's coordinates and re-read all of it)
            const uint32_t orient = (aff.y >> 24) & 3u;                                 // -(dir+1) mod 4 of the viewer
            const uint32_t slot = s_oslot[base];
            const bool stacked = show != 0xFF && slot != 0xFF;                          // an agent on an overlappable object / an empty cell
            const uint32_t sh = stacked ? show : 0u;
            const uint32_t sdir = w_recb[sh * 8 + MG_AG_DIR];
            uint32_t tile = 1 + base;
            if (stacked) tile = 1 + cfg.n_obj + (__umul24(slot, (uint32_t)n) + show) * 4 + sdir;
            uint32_t vt = __umul24(orient, (uint32_t)cfg.n_tiles) + tile;             // (virtual) tile index
            bool dyn = false;
            if constexpr (kPrestige) {
                if (stacked && ((cfg.prestige_mask >> show) & 1u) && (w_recb[show * 8 + MG_AG_FLAGS] & MG_AF_ACTIVE)) {
                    // hidden object under it: the plain-cell-object set
                    const uint32_t hv = (cfg.any_hide && base == 0 &&
                                         w_grid[__umul24(w_recb[show * 8 + MG_AG_X], (uint32_t)H) + w_recb[show * 8 + MG_AG_Y]] != 0) ? (uint32_t)n : 0u;
                    vt = NT4 + (hv + show) * 4 + orient;
                    dyn = true;
                }
            }
            const uint32_t iv = __umul24(v, (uint32_t)VV) + __umul24(vb, (uint32_t)VS) + va;
            if constexpr (kChunkRaster && !kGlobalAtlas)                                // dword offset from the atlas base
                tmap[iv] = (uint16_t)(dyn ? dyn_off / 4 + __umul24(vt - NT4, (uint32_t)(TS_ * TS_ * 3 / 4)) : __umul24(vt, (uint32_t)(TS_ * TS_ * 3 / 4)));
            else
                tmap[iv] = (uint16_t)vt;
            if (dbg_cells) {
                const size_t o = ((size_t)(e + (int)g) * nv + v) * VV + va * VS + vb;  // [i][j] like the reference
                dbg_cells[o] = (uint8_t)base;
                dbg_agent[o] = (uint8_t)show;
            }
            return see;
        };
        // Compile-time views: one lane per view ROW (G * nv * VS of them: 42 of 64 lanes for two envs of three 7 x 7 views, two
        // trips for four) — the row's origin and step worked out once (along a view row the world cell moves by one column or
        // one row: p = swap ? vb : va), its VS cells unrolled, the row's transparency mask assembled in a register and stored
        // once.  Round 5's form — a lane per CELL: three divisions, the viewer's map decoded and an LDS atomic per cell — stays
        // for run-time view sizes and for groups with fewer than 24 rows (one env of three viewers: 21 lanes walking 7 cells each
        // is a longer chain than 147 cells over 64 lanes — measured: BASELINE configs[1], one env per wave, +0.9 % in row form;
        // four envs of ONE viewer, 28 rows — the reference's human_player example —: -1.4 %).
        constexpr bool kRowViews = VS_ > 0;
        if (kRowViews && G * nv * VS >= 24) {
            const uint32_t rows = (uint32_t)(G * nv * VS);
            for (uint32_t it = (uint32_t)lane; it < rows; it += kWave) {
                const uint32_t gv = by_VS.template div<true>(it), vb = it - __umul24(gv, (uint32_t)VS);
                const uint32_t g = by_nv.div(gv), v = gv - __umul24(g, (uint32_t)nv);
                const uint8_t* w_grid = g_grid + __umul24(g, (uint32_t)cfg.cells_stride);
                const uint32_t gcell = __umul24(g, (uint32_t)L.cell_stride);
                const uint2 aff = w_vaff[__umul24(g, (uint32_t)nv) + v];
                const uint8_t* w_recb = reinterpret_cast<const uint8_t*>(g_rec + __umul24(g, (uint32_t)rec_stride));   // records, bytewise
                uint16_t* tmap = w_tmap + __umul24(g, (uint32_t)(L.tmap_stride / 2));
                const bool swap = (aff.x >> 20) & 1u, negx = (aff.x >> 21) & 1u, negy = (aff.x >> 22) & 1u;
                const int ox = (int)(aff.x & 0x3FFu) - 256, oy = (int)((aff.x >> 10) & 0x3FFu) - 256;
                // column va = 0 of the row, and the step to the next column
                int wx = ox + (swap ? (negx ? -(int)vb : (int)vb) : 0), wy = oy + (swap ? 0 : (negy ? -(int)vb : (int)vb));
                const int dx = swap ? 0 : (negx ? -1 : 1), dy = swap ? (negy ? -1 : 1) : 0;
                uint32_t bits = 0;
#pragma unroll
                for (int va = 0; va < (VS_ ? VS_ : 1); va++) {
                    if (view_cell(g, v, (uint32_t)va, vb, wx, wy, w_grid, gcell, aff, w_recb, tmap)) bits |= 1u << va;
                    wx += dx; wy += dy;
                }
                w_trow[__umul24(g, (uint32_t)L.trow_stride) + __umul24(v, (uint32_t)VS) + vb] = bits;
            }
        } else {
            for (uint32_t it = (uint32_t)lane; it < (uint32_t)(G * nvVV); it += kWave) {
                const uint32_t g = by_nvVV.div(it), iv = it - __umul24(g, (uint32_t)nvVV);
                const uint32_t v = by_VV.template div<kExactVV>(iv), c = iv - __umul24(v, (uint32_t)VV);
                const uint32_t vb = by_VS.template div<(VS_ > 0)>(c), va = c - __umul24(vb, (uint32_t)VS);
                const uint8_t* w_grid = g_grid + __umul24(g, (uint32_t)cfg.cells_stride);
                const uint32_t gcell = __umul24(g, (uint32_t)L.cell_stride);
                const uint2 aff = w_vaff[__umul24(g, (uint32_t)nv) + v];
                const bool swap = (aff.x >> 20) & 1u;
                const int p = (int)(swap ? vb : va), q = (int)(swap ? va : vb);
                const int wx = (int)(aff.x & 0x3FFu) - 256 + (((aff.x >> 21) & 1u) ? -p : p);
                const int wy = (int)((aff.x >> 10) & 0x3FFu) - 256 + (((aff.x >> 22) & 1u) ? -q : q);
                const uint8_t* w_recb = reinterpret_cast<const uint8_t*>(g_rec + __umul24(g, (uint32_t)rec_stride));   // records, bytewise
                uint16_t* tmap = w_tmap + __umul24(g, (uint32_t)(L.tmap_stride / 2));
                if (view_cell(g, v, va, vb, wx, wy, w_grid, gcell, aff, w_recb, tmap))
                    atomicOr(&w_trow[__umul24(g, (uint32_t)L.trow_stride) + __umul24(v, (uint32_t)VS) + vb], 1u << va);
            }
        }
        wave_lds_sync();
        // 4. visibility, one lane per viewer
        auto visibility = [&](const int g, const int v) {         // slot g, viewer v
            const uint64_t r = g_rec[__mul24(g, rec_stride) + s_vmap[v]];
            const int row0 = __mul24(g, L.trow_stride) + __mul24(v, VS);
            if (VS_ == 0 && VS > kRegView) {                            // views of 16 .. 31 rows: the rows stay in LDS
                uint32_t* w_m = reinterpret_cast<uint32_t*>(ws + L.trow2) + row0;
                if (!(rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE)) { for (int j = 0; j < VS; j++) w_m[j] = 0; }
                else if (cfg.see_through_walls) { for (int j = 0; j < VS; j++) w_m[j] = (1u << VS) - 1u; }
                else occlude_rows_mem(VS, off, &w_trow[row0], w_m);
                for (int j = 0; j < VS; j++) w_vis[row0 + j] = w_m[j];
                return;
            }
            uint32_t m[VS_ ? VS_ : kRegView];
            if (!(rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE)) {           // base.py:420-425
                for (int j = 0; j < VS; j++) m[j] = 0;
            } else if (cfg.see_through_walls) {                          // agents.py:294-295
                for (int j = 0; j < VS; j++) m[j] = (1u << VS) - 1u;
            } else {
                occlude_rows<VS_>(VS, off, &w_trow[row0], m);
            }
            for (int j = 0; j < VS; j++) w_vis[row0 + j] = m[j];
        };
        for (int it = lane; it < G * nv; it += kWave) { const int g = (int)by_nv.div((uint32_t)it); visibility(g, it - __mul24(g, nv)); }
        wave_lds_sync();
        // 5. shadow (base.py:313-316): the cells a viewer does not see take tile 0 — the same pixels in every
        //    orientation, offset 0 of the atlas.  One lane per view ROW: its visibility mask, then its VS cells.
        {
            const uint32_t nvVS = (uint32_t)(nv * VS);
            const Div20 by_nvVS(nvVS, lc.m_nvVS);
            for (uint32_t it = (uint32_t)lane; it < (uint32_t)G * nvVS; it += kWave) {
                const uint32_t g = by_nvVS.div(it), r = it - __umul24(g, nvVS);          // r = viewer * VS + view row
                const uint32_t mask = w_vis[__umul24(g, (uint32_t)L.trow_stride) + r];
                uint16_t* row = w_tmap + __umul24(g, (uint32_t)(L.tmap_stride / 2)) + __umul24(r, (uint32_t)VS);
                for (int va = 0; va < VS; va++)
                    if (!((mask >> va) & 1u)) row[va] = 0;
                if (dbg_cells) {
                    const uint32_t v = by_VS.template div<(VS_ > 0)>(r), vb = r - __umul24(v, (uint32_t)VS);
                    for (int va = 0; va < VS; va++)
                        dbg_vis[((size_t)(e + (int)g) * nv + v) * VV + va * VS + vb] = (uint8_t)((mask >> va) & 1u);
                }
            }
        }
        wave_lds_sync();
        }
        }
        wave_lds_sync();
        } else {
        if constexpr (kPrestige) {
            // 4b. tiles of active 'prestige' agents are recoloured per env (render_post) — and blended
            //     with the object they stand on — before rotation, for all 4 orientations.  With
            //     hide_item_types a viewer may hide that object and see the agent as a plain cell
            //     object instead: a second set (hv = 1) on the empty tile.
            //     Colours first, one lane per agent (the float64 tanh runs once per env, not once per
            //     agent); w_trow is free after the shadow cast.  Then per agent the tile in orientation
            //     0 (the only pass with per-pixel arithmetic) and three rotated byte copies of it.
            const int npx = TS * TS;
            // byte offset of pixel p inside a tile, and the bytes of a tile: as in HBM, or — gather raster — in padded rows
            auto px = [&](int p) -> int {
                if constexpr (kGather) { const int r = p / TS_, c = p - r * TS_; return r * Gm::RS + Gm::FRONT + 3 * c; }
                else return 3 * p;
            };
            const int tbytes = kGather ? Gm::TILE : tile_bytes;
            const uint8_t* abase = kGlobalAtlas ? cfg.atlas : s_atlas;
            // (a grid that is read in place: the object a 'prestige' agent stands on comes from where the grid lives)
            const uint8_t* w_grid = kBigGrid ? kernarg_again<MgState>(offsetof(RenderKernargs, st)).grid + (size_t)e * cfg.cells_stride
                                             : w_stage_g + (size_t)ej * cfg.cells_stride;      // (here, per env: the recoloured tiles have ONE slot;
            const uint64_t* w_rec = w_stage_r + (size_t)ej * rec_stride;            //  the views of the group are done, w_trow is free)
            uint32_t* w_col = w_trow;
            if (lane < n && ((cfg.prestige_mask >> lane) & 1u)) {
                if (fs.enabled) w_col[lane] = w_stage_c[(size_t)ej * rec_stride + lane];   // (computed by the env's stepping lane)
                else {
                    const PrestigeColor c = prestige_color(w_stage_p[(size_t)ej * rec_stride + lane], s_pscale[lane]);
                    w_col[lane] = c.r | (c.g << 8) | (c.b << 16);
                }
            }
            wave_lds_sync();
            // (an agent's tile is looked at in the orientation of the VIEWER — phase 5 —: only those of this env's
            // viewers are made, one of the four for a single agent)
            uint32_t omask = 0;
            for (int v = 0; v < nv; v++) omask |= 1u << ((w_vaff[(ej - ej0) * nv + v].y >> 24) & 3u);
            for (int Xh = 0; Xh < (cfg.any_hide ? 2 * n : n); Xh++) {
                const int hv = Xh >= n, X = Xh - hv * n;
                if (!((cfg.prestige_mask >> X) & 1u)) continue;
                const uint64_t rx = w_rec[X];
                if ((rec_byte(rx, MG_AG_FLAGS) & (MG_AF_ACTIVE | MG_AF_PLACED)) != (MG_AF_ACTIVE | MG_AF_PLACED)) continue;
                uint32_t base = w_grid[rec_byte(rx, MG_AG_X) * H + rec_byte(rx, MG_AG_Y)];
                if (hv) { if (!base) continue; base = 0; }
                const uint32_t sdir = rec_byte(rx, MG_AG_DIR);
                const uint32_t pc = w_col[X];
                const PrestigeColor col = {pc & 0xFFu, (pc >> 8) & 0xFFu, (pc >> 16) & 0xFFu};
                const uint32_t amax4 = (uint32_t)cfg.prestige_amax[0] | ((uint32_t)cfg.prestige_amax[1] << 8) |
                                       ((uint32_t)cfg.prestige_amax[2] << 16) | ((uint32_t)cfg.prestige_amax[3] << 24);
                const uint32_t amax = (amax4 >> (8u * sdir)) & 0xFFu;   // (no indexed kernarg load: that is a VMEM load)
                const uint32_t M = ((amax * col.r) >> 8) + ((amax * col.g) >> 8) + ((amax * col.b) >> 8);
                const uint8_t* white = abase + (size_t)(cfg.prestige_sprite_tile + sdir) * tbytes;   // orientation 0, no border
                const uint8_t* btile = base ? abase + (size_t)(1 + base) * tbytes : nullptr;
                const bool border = (s_oflags2[base] & 1) != 0;
                const uint8_t* etile = abase + (size_t)tbytes;                                   // empty tile
                uint8_t* t0 = w_dyn + (size_t)(Xh * 4) * tbytes;
                for (int p = lane; p < npx; p += kWave) {
                    const int sp = px(p);
                    prestige_pixel(white[sp], col, M, btile ? btile + sp : nullptr, border ? etile + sp : nullptr, t0 + sp);
                }
                wave_lds_sync();
                for (int o = 1; o < 4; o++) {
                    if (!((omask >> o) & 1u)) continue;
                    for (int p = lane; p < npx; p += kWave) {
                        const int r = p / TS, c = p - r * TS;
                        int sr, sc;   // source pixel of output pixel (r, c) at orientation o (rotate_grid, base.py:67-80)
                        if (o == 3) { sr = c; sc = TS - 1 - r; }
                        else if (o == 1) { sr = TS - 1 - c; sc = r; }
                        else { sr = TS - 1 - r; sc = TS - 1 - c; }
                        const uint8_t* src = t0 + px(sr * TS + sc);
                        uint8_t* dst = t0 + (size_t)o * tbytes + px(p);
                        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
                    }
                }
            }
            wave_lds_sync();
        }
        // 6. raster: stream the env's n images out
        // The rasters that are bound by instruction issue run at a raised wave priority: a wave that is storing gets its
        // instructions before the SIMD's other waves' step / view instructions, so the launch's stores start earlier and stop
        // less often (whole step: tile 5 -4.2 %, tile 6 -1.2 %, view 9 at tile 5 -3.8 %, tile 13 — assemble-and-stream — -4.7 %,
        // the reference's example -1.8 %; tile 8, HBM-bound: +0.01 %, tile 11: +0.6 % — both left alone; profiles/r05).
        if constexpr (kRasterPrio) __builtin_amdgcn_s_setprio(3);
        if constexpr (kGather) {
            // the whole GROUP's images as one stream (mg_gather.h), when the group's first env comes up — or, 'prestige', env
            // by env: the recoloured tiles (w_dyn: ONE slot, virtual tiles >= NT4) are this env's
            if constexpr (kPrestige) {
                gather_group<VS_, TS_, true>(lane, reinterpret_cast<const uint8_t*>(w_tmap), s_atlas, obs + (size_t)e * nv * img_bytes,
                                             (uint32_t)(nv * img_bytes), GatherDyn{NT4, dyn_off - NT4 * (uint32_t)Gm::TILE});
            } else if (ej == ej0) {
                const int G = min(kb, ej0 + gd) - ej0;
                gather_group<VS_, TS_>(lane, reinterpret_cast<const uint8_t*>(w_tmap0), s_atlas, obs + (size_t)e * nv * img_bytes,
                                       (uint32_t)((size_t)G * nv * img_bytes));
            }
        } else if constexpr (kChunkRaster) {
            // The env's n images are one contiguous run of 8-byte *pairs*: PR pairs per pixel row,
            // PT per tile row (TD even => a pair never straddles a tile row, and every pair is
            // 8-byte aligned in the atlas: ds_read_b64).  A 16-byte chunk is pairs (2c, 2c+1).
            // tmap is laid out [image][band][column], and an image has exactly VS bands, so the
            // GLOBAL pixel row r (counted across the env's images) indexes it directly:
            // tile = tmap[(r / TS) * VS + column].  A lane's chunk advances by 64 chunks = 128
            // pairs per trip, so (r, pair-in-row) is carried incrementally: no per-chunk division
            // by anything but the compile-time PT / TS (24-bit multiply-shift, all values < 2^16).
            constexpr uint32_t TD = TS_ * 3 / 4, PT = TD / 2;     // dwords / pairs per tile row
            constexpr uint32_t M_PT = (65536u + PT - 1) / PT, M_TS = (65536u + TS_ - 1) / TS_;
            const uint32_t* atlas32 = reinterpret_cast<const uint32_t*>(s_atlas);
            const uint32_t* gatlas32 = reinterpret_cast<const uint32_t*>(cfg.atlas);
            auto ld_pair = [&](uint32_t a) -> uint2 {
                if constexpr (kSplit) {
                    if (a & kInLds) return *reinterpret_cast<const uint2*>(atlas32 + (a & ~kInLds));
                    return *reinterpret_cast<const uint2*>(gatlas32 + a);
                } else if constexpr (kGlobalAtlas) return *reinterpret_cast<const uint2*>(gatlas32 + a);
                else return *reinterpret_cast<const uint2*>(atlas32 + a);
            };
            uint4* out = reinterpret_cast<uint4*>(obs + (size_t)e * nv * img_bytes);
            const int total = (int)(nv * (img_bytes / 16));
            uint32_t r = rast_r0, pr = rast_p0;
            // pair (global pixel row rr_, pair-in-row pr_) -> its tmap index and its dword offset inside the tile
            auto pair_coords = [&](uint32_t rr_, uint32_t pr_, uint32_t& ti, uint32_t& of) {
                // va = pr_ / PT and kp = pr_ % PT from ONE product: its high half is the quotient, its low half times PT
                // the remainder.  (Written as pr_ - va * PT, the compiler fuses the multiply into a v_mad_u64_u32 on
                // the (pr, r) register pair it keeps for the carry trick of the advance — a quarter-rate instruction,
                // eight per trip.)
                const uint32_t prod = __umul24(pr_, M_PT);
                const uint32_t va = prod >> 16, kp = __umul24(prod & 0xFFFFu, PT) >> 16;
                uint32_t vb, rr;
                if constexpr ((TS_ & (TS_ - 1)) == 0) { vb = rr_ / (uint32_t)TS_; rr = rr_ & (uint32_t)(TS_ - 1); }   // (shifts)
                else { vb = __umul24(rr_, M_TS) >> 16; rr = rr_ - __umul24(vb, (uint32_t)TS_); }
                ti = __umul24(vb, (uint32_t)VS) + va;
                of = __umul24(rr, TD) + kp * 2u;
            };
            auto tile_dword = [&](uint32_t t) -> uint32_t {      // tmap entry -> dword offset of the tile
                if constexpr (kSplit)
                    return t < NT4 ? t * (uint32_t)(TS_ * TS_ * 3 / 4)
                                   : (kInLds | (dyn_off / 4 + (t - NT4) * (uint32_t)(TS_ * TS_ * 3 / 4)));
                else if constexpr (kGlobalAtlas) return t * (uint32_t)(TS_ * TS_ * 3 / 4);   // tile index -> dword offset
                else return t;
            };
            auto pair_addr = [&](uint32_t rr_, uint32_t pr_) -> uint32_t {
                uint32_t ti, of;
                pair_coords(rr_, pr_, ti, of);
                return tile_dword((uint32_t)w_tmap[ti]) + of;
            };
            auto fetch = [&](uint4& v) {
                if constexpr (V_ == 4) {
                    v = make_uint4(0x1e19231eu, 0x231e1923u, 0x1e19231eu, 0x231e1923u);
                } else {
                    uint32_t r1 = r, pr1 = pr + 1;
                    if (pr1 == PR) { pr1 = 0; r1++; }
                    const uint2 p0 = ld_pair(pair_addr(r, pr));
                    const uint2 p1 = ld_pair(pair_addr(r1, pr1));
                    v = make_uint4(p0.x, p0.y, p1.x, p1.y);
                }
                pr += STEP_P; r += STEP_R;
                if (pr >= PR) { pr -= PR; r++; }
            };
            auto put = [&](int c, const uint4& v) {
                if constexpr (V_ == 2) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 nv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4*>(out + c));
                } else out[c] = v;
            };
            // FIXED-LANE mapping (compile-time view size, tile 8 / 16): the env's stream repeats every PRW pixel rows
            // = PC 16-byte chunks (21 for view 7: two rows of 168 B at tile 8, one row of 336 B at tile 16), so a
            // wave-instruction that covers a whole number of periods — 63 lanes = 3 periods, 1 008 contiguous bytes;
            // lane 63 idles — gives every lane the SAME place in the period on every trip: which pixel row of the
            // period, which view column, which pair of the tile row are per-lane constants, and a trip only
            // advances the row by a constant.  Per pair and trip: shift, and, two 24-bit multiply-adds — where the
            // general mapping (64 chunks per trip, the place in the period rotating) re-derives column and pair
            // with a division each time: 157 VALU instructions per 4 KiB against ~70.  (The chunk raster is not
            // purely HBM-bound: SQ counters show its VALU pipes 60 % busy, profiles/r03.)
            constexpr uint32_t RBc = (uint32_t)(VS_ ? VS_ : 1) * TS_ * 3;                 // bytes per pixel row
            constexpr uint32_t PRW = (RBc % 16u) ? 2u : 1u, PC = PRW * RBc / 16u;          // rows / chunks per period
            constexpr uint32_t GPT = PC ? 64u / PC : 0u, LU = GPT * PC;                    // periods / lanes used per trip
            constexpr bool kFixedLane = VS_ > 0 && LU >= 60 && V_ != 6;
            if constexpr (kFixedLane) {
                constexpr uint32_t PRc = RBc / 8u;                                          // pairs per pixel row
                const uint32_t q = (uint32_t)lane % PC, grp = (uint32_t)lane / PC;
                uint32_t rowpar[2], vaq[2], kp2[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t pj = 2u * q + (uint32_t)j;                               // pair within the period
                    rowpar[j] = pj / PRc;
                    const uint32_t prj = pj - rowpar[j] * PRc;
                    vaq[j] = prj / PT;
                    kp2[j] = (prj - vaq[j] * PT) * 2u;
                }
                const bool live = (uint32_t)lane < LU;
                uint32_t rcur = PRW * grp;                                                  // first row of this lane's period
                auto coords = [&](uint32_t rr_, int j, uint32_t& ti, uint32_t& of) {
                    uint32_t vb, rr;
                    if constexpr ((TS_ & (TS_ - 1)) == 0) { vb = rr_ / (uint32_t)TS_; rr = rr_ & (uint32_t)(TS_ - 1); }
                    else { vb = __umul24(rr_, M_TS) >> 16; rr = rr_ - __umul24(vb, (uint32_t)TS_); }
                    ti = __umul24(vb, (uint32_t)VS) + vaq[j];
                    of = __umul24(rr, TD) + kp2[j];
                };
                constexpr int CSF = (int)LU;
                int base = 0;                                                               // first chunk of the trip (uniform)
                for (; base + 4 * CSF <= total; base += 4 * CSF) {                          // four WHOLE trips at a time
                    if (live) {                                                             // (one exec mask for all of it)
                        const int c = base + lane;
                        uint32_t ti[8], of[8];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            coords(rcur + rowpar[0], 0, ti[2 * t], of[2 * t]);
                            coords(rcur + rowpar[1], 1, ti[2 * t + 1], of[2 * t + 1]);
                            rcur += PRW * GPT;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        uint2 pp[8];
                        if constexpr (V_ != 4) {
                            uint32_t tt[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) tt[i] = (uint32_t)w_tmap[ti[i]];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < 8; i++) pp[i] = ld_pair(tile_dword(tt[i]) + of[i]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; i++) pp[i] = make_uint2(0x1e19231eu, 0x231e1923u);
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++) put(c + t * CSF, make_uint4(pp[2 * t].x, pp[2 * t].y, pp[2 * t + 1].x, pp[2 * t + 1].y));
                    }
                }
                for (; base < total; base += CSF) {                                         // the remaining trips, one by one
                    const int c = base + lane;
                    uint32_t ti0, of0, ti1, of1;
                    coords(rcur + rowpar[0], 0, ti0, of0);
                    coords(rcur + rowpar[1], 1, ti1, of1);
                    rcur += PRW * GPT;
                    if (live && c < total) {
                        uint2 p0, p1;
                        if constexpr (V_ != 4) { p0 = ld_pair(tile_dword((uint32_t)w_tmap[ti0]) + of0); p1 = ld_pair(tile_dword((uint32_t)w_tmap[ti1]) + of1); }
                        else { p0 = make_uint2(0x1e19231eu, 0x231e1923u); p1 = p0; }
                        put(c, make_uint4(p0.x, p0.y, p1.x, p1.y));
                    }
                }
            } else {
            int c = (int)c_first;
            constexpr int CS = (int)CH_STRIDE;
            if constexpr (V_ == 4) {
                for (; c + 3 * CS < total; c += 4 * CS) {
                    uint4 v0, v1, v2, v3;
                    fetch(v0); fetch(v1); fetch(v2); fetch(v3);
                    put(c, v0); put(c + CS, v1); put(c + 2 * CS, v2); put(c + 3 * CS, v3);
                }
            } else if constexpr (V_ != 6) {
                // Four chunks per trip = eight pairs, in THREE phases with nothing scheduled across them: the eight
                // tmap look-ups issued together (one LDS round trip), then the eight atlas reads (a second), then
                // the four 1-KiB stores back to back.  Written pair by pair, the compiler — at the 128-VGPR limit
                // of the 16-wave instantiations — serialises it into twelve dependent LDS round trips per trip.
                for (; c + 3 * CS < total; c += 4 * CS) {
                    uint32_t ti[8], of[8];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        uint32_t r1 = r, pr1 = pr + 1;
                        if (pr1 == PR) { pr1 = 0; r1++; }
                        pair_coords(r, pr, ti[2 * q], of[2 * q]);
                        pair_coords(r1, pr1, ti[2 * q + 1], of[2 * q + 1]);
                        pr += STEP_P; r += STEP_R;
                        if (pr >= PR) { pr -= PR; r++; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    uint32_t tt[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) tt[i] = (uint32_t)w_tmap[ti[i]];
                    __builtin_amdgcn_sched_barrier(0);
                    uint2 pp[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) pp[i] = ld_pair(tile_dword(tt[i]) + of[i]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; q++) put(c + q * CS, make_uint4(pp[2 * q].x, pp[2 * q].y, pp[2 * q + 1].x, pp[2 * q + 1].y));
                }
            }
            for (; c < total; c += CS) {
                uint4 v;
                fetch(v);
                put(c, v);
            }
            }
        } else {
            // Any tile size: the env's n*P*P*3 output bytes are the next S bytes of the wave's stream;
            // tile rows are SEG = 3*TS bytes at arbitrary byte offsets.  Piece by piece (piece_rows whole
            // pixel rows, ~4 KiB): ASSEMBLE — segment g of the piece (pixel row g / VS, view column
            // g % VS) is SEG contiguous bytes of one atlas tile row, ORed by one lane into the zeroed
            // w_out at carry + g*SEG; then STREAM
            // the complete 16-byte chunks of w_out (linear ds_read_b128 -> global_store_dwordx4) and move
            // the incomplete tail to the front, where the next piece — or the next env — continues.
            const uint32_t SEG = 3u * (uint32_t)TS, P = (uint32_t)(VS * TS), RB = P * 3u;
            const uint32_t NR = (uint32_t)nv * P;                          // pixel rows per env
            const Div20 by_TS((uint32_t)TS);
            // (compile-time tile size: R < 16 viewers * 15 * 64 pixel rows, and R * (m * TS - 2^20) < 2^20 then)
            constexpr bool kExactTS = TS_ > 0 && (16u * 15u * TS_ * (((((1u << 20) + (TS_ ? TS_ : 1) - 1u) / (TS_ ? TS_ : 1)) * (TS_ ? TS_ : 1)) - (1u << 20)) < (1u << 20));
            auto tile_off = [&](uint32_t vt) -> uint32_t {                 // virtual tile index -> byte offset
                if constexpr (kSplit) return vt < NT4 ? __umul24(vt, (uint32_t)tile_bytes) : (kInLds | (dyn_off + __umul24(vt - NT4, (uint32_t)tile_bytes)));
                else if constexpr (kPrestige) return vt < NT4 ? __umul24(vt, (uint32_t)tile_bytes) : dyn_off + __umul24(vt - NT4, (uint32_t)tile_bytes);
                else return __umul24(vt, (uint32_t)tile_bytes);
            };
            for (uint32_t R0 = 0; R0 < NR; R0 += (uint32_t)L.piece_rows) {
                const uint32_t rows = min((uint32_t)L.piece_rows, NR - R0), nseg = rows * (uint32_t)VS;
                for (uint32_t g = lane; g < nseg; g += kWave) {
                    const uint32_t Rl = by_VS.div(g), col = g - __umul24(Rl, (uint32_t)VS), R = R0 + Rl;
                    // band = R / TS and rr = R % TS from ONE 24-bit product when the quotient is exact (R - band * TS
                    // is fused into a quarter-rate v_mad_u64_u32 otherwise): the fraction times TS, rounded down
                    uint32_t band, rr;
                    if constexpr (kExactTS) {
                        const uint32_t prod = __umul24(R, by_TS.m);
                        band = prod >> 20;
                        rr = __umul24(prod & 0xFFFFFu, (uint32_t)TS) >> 20;
                    } else {
                        band = by_TS.template div<false>(R);
                        rr = R - __umul24(band, (uint32_t)TS);
                    }
                    const uint32_t so = tile_off((uint32_t)w_tmap[__umul24(band, (uint32_t)VS) + col]) + __umul24(rr, SEG);
                    const uint8_t* sb;       // source base the offset `sa` counts from
                    uint32_t sa = so;
                    if constexpr (kSplit) { sb = (so & kInLds) ? s_atlas : cfg.atlas; sa = so & ~kInLds; }
                    else if constexpr (kGlobalAtlas) sb = cfg.atlas;
                    else sb = s_atlas;
                    or_segment<TS_ * 3>(sb, sa, w_out, carry + __umul24(g, SEG), SEG);
                }
                wave_lds_sync();
                const uint32_t total = carry + rows * RB, full = total >> 4;
                if (full) {
                    uint32_t c0 = 0;
                    if (head) {   // chunk 0 is shared with the wave before this one: only our bytes of it
                        if ((uint32_t)lane >= head && lane < 16) out_base[lane] = w_out[lane];
                        head = 0;
                        c0 = 1;
                    }
                    const uint4* src16 = reinterpret_cast<const uint4*>(w_out);
                    uint4* dst16 = reinterpret_cast<uint4*>(out_base);
                    uint32_t c = c0 + (uint32_t)lane;
                    for (; c + 3u * kWave < full; c += 4u * kWave) {
                        const uint4 v0 = src16[c], v1 = src16[c + kWave], v2 = src16[c + 2 * kWave], v3 = src16[c + 3 * kWave];
                        dst16[c] = v0; dst16[c + kWave] = v1; dst16[c + 2 * kWave] = v2; dst16[c + 3 * kWave] = v3;
                    }
                    for (; c < full; c += kWave) dst16[c] = src16[c];
                    const uint32_t tail = total - (full << 4);
                    const uint8_t tb = (uint32_t)lane < tail ? w_out[(full << 4) + lane] : (uint8_t)0;
                    wave_lds_sync();
                    // the incomplete chunk moves to the front (zero-padded to 16 bytes); the rest of the buffer is
                    // zeroed for the next piece's ORs
                    if (lane < 16) w_out[lane] = tb;
                    for (int i = 1 + lane; i < L.out_chunks; i += kWave) reinterpret_cast<uint4*>(w_out)[i] = make_uint4(0, 0, 0, 0);
                    out_base += (size_t)full << 4;
                    carry = tail;
                } else {
                    carry = total;
                }
                wave_lds_sync();
            }
            if (e + 1 == e_end && carry > head) {   // end of the run: the bytes of the last, incomplete chunk
                if ((uint32_t)lane >= head && (uint32_t)lane < carry) out_base[lane] = w_out[lane];
            }
        }
        }
        if constexpr (kRasterPrio) __builtin_amdgcn_s_setprio(0);
        wave_lds_sync();   // scratch is reused by the next env
        if (e == e0) MG_STAMP(pass == 0 ? 4 : 5);
    }
        if constexpr (kEnc) { if (enc_pending) encode_batch(false); }
    }
    MG_STAMP(6);
}


// Compute units of the current device (256 on a whole MI355X; 32 per partition in CPX mode): what the persistent
// grid is sized for.  Asked once per device (hipDeviceGetAttribute is a host-side table look-up, no stream work).
inline int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

template <int VS_, int TS_, int WPB, int VX_ = 0, int RM_ = 0>
hipError_t launch_render_t(const MgConfig& cfg, const MgState& st, uint8_t* obs, uint8_t* c, uint8_t* a,
                                  uint8_t* v, hipStream_t s, const FusedStep* fs, RenderPick* pick) {
    static_assert(RM_ == 1 || TS_ == 0 || (TS_ % 8) != 0 || TS_ == 8 || TS_ == 16 || TS_ == 32, "see render_chunk_raster");
    const RenderScratch L = render_scratch_for(cfg, WPB, RM_);
    constexpr int V_ = VX_ & 15;
    const size_t atlas_lds = (V_ == 8 || V_ == 12) ? 0 : (size_t)render_atlas_lds_bytes(cfg, RM_);
    const RenderShared sh = render_shared_layout(cfg);
    // (mg_step_render_encode: its table — fs->enc_ne dwords — lies behind the block-shared tables, the waves' scratch behind it)
    constexpr bool kEnc = (VX_ & 16) != 0;
    if (kEnc && (!fs || fs->enc_ne <= 0)) return hipErrorInvalidValue;
    const size_t enc_lds = kEnc ? (size_t)fs->enc_ne * 4 : 0;
    size_t lds = atlas_lds + sh.total + enc_lds + WPB * (size_t)L.total;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (pick) {     // mg_render_kernel_name: which instantiation this configuration gets — nothing is launched
        pick->vs = VS_; pick->ts = TS_; pick->wpb = WPB; pick->v = VX_; pick->rm = RM_; pick->lds = (int)lds;
        return hipSuccess;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<VS_, TS_, WPB, VX_, RM_>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // Persistent grid: at most the workgroups that are co-resident on the device's CUs (each stages the atlas
    // once), sized so that every wave walks the same number of envs (an uneven tail costs up to one
    // env-time in ~6).  Registers (~87 VGPRs) admit 5 waves per SIMD = 20 per CU.
    int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > 20 / WPB) per_cu = 20 / WPB;
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_RENDER_PER_CU")) { const int v = atoi(f); if (v >= 1 && v < per_cu) per_cu = v; }
#endif
    if (per_cu < 1) per_cu = 1;
    const int max_blocks = device_cus() * per_cu;
    const int need = (cfg.B + WPB - 1) / WPB;   // workgroups if every wave took one env
    const int rounds = (need + max_blocks - 1) / max_blocks;
    int blocks = (need + rounds - 1) / rounds;
#if defined(MG_AB_VARIANTS)
    // measurement build (tools/ab_rounds.py): more workgroups than are resident at a time — the later ones start as the
    // first ones exit, their store-free heads under the others' stores (VERDICT r02 item 4a within ONE launch)
    if (const char* f = getenv("MG_RENDER_OVERSUB")) { const int v = atoi(f); if (v > 1) blocks = min(need, blocks * v); }
#endif
    RenderLaunch lc;
    lc.L = L;
    lc.per_wave = (cfg.B + blocks * WPB - 1) / (blocks * WPB);
    const uint32_t nv = (uint32_t)(cfg.n_view ? cfg.n_view : cfg.n_agents), vs = (uint32_t)cfg.view_size;
    lc.m_n = Div20((uint32_t)cfg.n_agents).m;
    lc.m_nv = Div20(nv).m;
    lc.m_nvVV = Div20(nv * vs * vs).m;
    lc.m_VV = Div20(vs * vs).m;
    lc.m_VS = Div20(vs).m;
    lc.m_nvVS = Div20(nv * vs).m;
    lc.depth_mode = 0;
    lc.atlas_lds = (int)atlas_lds;
    lc.sh = sh;
#if defined(MG_AB_VARIANTS)
    lc.stamps = g_ab_stamps;
#endif
#if defined(MG_AB_VARIANTS)
    if (const char* f = getenv("MG_RENDER_DEPTH")) lc.depth_mode = atoi(f);   // 1: every wave view -> raster env by env
#endif
    hipLaunchKernelGGL((render_kernel<VS_, TS_, WPB, VX_, RM_>), dim3(blocks), dim3(WPB * 64), lds, s, cfg, st, obs, c, a, v,
                       lc, *fs);
    return hipGetLastError();
}

// The instantiations, in groups: one translation unit per group (mg_render_inst_<g>.hip) makes them — in parallel —,
// the dispatcher (mg_render.hip) only refers to them.  MG_RENDER_GROUP_x(X): X(VS, TS, WPB, V, RM).
#define MG_RENDER_GROUP_A(X) /* the chunk raster at tile 8 */                                                              \
    X(7, 8, 16, 0, 0) X(7, 8, 4, 0, 0) X(9, 8, 16, 0, 0) X(9, 8, 4, 0, 0) X(5, 8, 16, 0, 0) X(5, 8, 4, 0, 0)   \
    X(3, 8, 16, 0, 0) X(3, 8, 4, 0, 0) X(0, 8, 8, 0, 0) X(0, 8, 4, 0, 0)
#define MG_RENDER_GROUP_B(X) /* tile 16 / 32, the atlas in global memory */                                                \
    X(7, 16, 16, 0, 0) X(7, 16, 4, 0, 0) X(7, 32, 16, 0, 0) X(7, 32, 4, 0, 0) X(0, 16, 8, 0, 0) X(0, 16, 4, 0, 0)               \
    X(0, 32, 8, 0, 0) X(0, 32, 4, 0, 0) X(0, 8, 4, 8, 0) X(0, 16, 4, 8, 0) X(0, 32, 4, 8, 0) X(0, 0, 4, 8, 0) X(0, 0, 4, 8, 3)
#define MG_RENDER_GROUP_C(X) /* assemble-and-stream: any other tile size */                                                \
    X(7, 0, 16, 0, 0) X(7, 0, 4, 0, 0) X(0, 0, 8, 0, 0) X(0, 0, 4, 0, 0)
#define MG_RENDER_GROUP_D(X) /* 'prestige': per-env recoloured tiles */                                                     \
    X(7, 8, 12, 9, 0) X(7, 8, 8, 9, 0) X(7, 8, 4, 9, 0)                                                                       \
    X(0, 8, 4, 9, 0) X(0, 16, 4, 9, 0)
#define MG_RENDER_GROUP_E(X)                                                                                               \
    X(7, 0, 12, 9, 0) X(7, 0, 8, 9, 0) X(7, 0, 4, 9, 0) X(0, 32, 4, 9, 0) X(0, 0, 4, 9, 0)                                      \
    X(0, 8, 4, 12, 0) X(0, 16, 4, 12, 0) X(0, 32, 4, 12, 0) X(0, 0, 4, 12, 0) X(0, 0, 4, 12, 3)
#if defined(MG_EXP) && (MG_EXP & 8)
#define MG_RENDER_GROUP_X(X) X(7, 8, 12, 0, 0)      /* experiment builds only (mg_render.hip) */
#else
#define MG_RENDER_GROUP_X(X)
#endif
#define MG_RENDER_GROUP_G(X) /* the gather raster (mg_gather.h): view 7, 5- and 6-pixel tiles */                          \
    X(7, 5, 16, 0, 2) X(7, 5, 4, 0, 2) X(7, 6, 16, 0, 2) X(7, 6, 4, 0, 2)
#define MG_RENDER_GROUP_H(X) /* ... 7-, 9- and 10-pixel tiles */                                                           \
    X(7, 7, 16, 0, 2) X(7, 7, 4, 0, 2) X(7, 9, 16, 0, 2) X(7, 9, 4, 0, 2) X(7, 10, 16, 0, 2) X(7, 10, 4, 0, 2)
#define MG_RENDER_GROUP_M(X) /* the gather raster for views 11, 13, 15 at 5-pixel tiles (8-wave workgroups) */                  \
    X(11, 5, 8, 0, 2) X(11, 5, 4, 0, 2) X(13, 5, 8, 0, 2) X(13, 5, 4, 0, 2) X(15, 5, 8, 0, 2) X(15, 5, 4, 0, 2)
#define MG_RENDER_GROUP_L(X) /* assemble-and-stream with a compile-time view: views 3 .. 9 at any tile size */                    \
    X(3, 0, 16, 0, 0) X(3, 0, 4, 0, 0) X(4, 0, 16, 0, 0) X(4, 0, 4, 0, 0) X(5, 0, 16, 0, 0) X(5, 0, 4, 0, 0)                       \
    X(6, 0, 16, 0, 0) X(6, 0, 4, 0, 0) X(8, 0, 16, 0, 0) X(8, 0, 4, 0, 0) X(9, 0, 16, 0, 0) X(9, 0, 4, 0, 0)
#define MG_RENDER_GROUP_K(X) /* the chunk raster at tile 8 for even views; the gather raster with 'prestige' agents at tile 5 */ \
    X(4, 8, 16, 0, 0) X(4, 8, 4, 0, 0) X(6, 8, 16, 0, 0) X(6, 8, 4, 0, 0) X(8, 8, 16, 0, 0) X(8, 8, 4, 0, 0)                       \
    X(7, 5, 12, 9, 2) X(7, 5, 8, 9, 2) X(7, 5, 4, 9, 2)
#define MG_RENDER_GROUP_J(X) /* ... views 3, 4, 5, 6, 8, 9 at 5-pixel tiles */                                                       \
    X(3, 5, 16, 0, 2) X(3, 5, 4, 0, 2) X(5, 5, 16, 0, 2) X(5, 5, 4, 0, 2) X(9, 5, 16, 0, 2) X(9, 5, 4, 0, 2)                       \
    X(4, 5, 16, 0, 2) X(4, 5, 4, 0, 2) X(6, 5, 16, 0, 2) X(6, 5, 4, 0, 2) X(8, 5, 16, 0, 2) X(8, 5, 4, 0, 2)
#define MG_RENDER_GROUP_I(X) /* ... 11- and 12-pixel tiles; 11 with 'prestige' agents (examples/human_player.py) */        \
    X(7, 11, 16, 0, 2) X(7, 11, 4, 0, 2) X(7, 12, 16, 0, 2) X(7, 12, 4, 0, 2) X(7, 11, 12, 9, 2) X(7, 11, 8, 9, 2) X(7, 11, 4, 9, 2)
#if defined(MG_AB_VARIANTS)
#define MG_RENDER_GROUP_V(X) /* measurement variants (tools/ab_render.py) */                                               \
    X(7, 8, 16, 2, 0) X(7, 8, 4, 2, 0) X(7, 8, 16, 3, 0) X(7, 8, 4, 3, 0) X(7, 8, 16, 4, 0) X(7, 8, 4, 4, 0)                     \
    X(7, 8, 16, 6, 0) X(7, 8, 4, 6, 0) X(7, 8, 16, 11, 0) X(7, 8, 4, 11, 0) X(7, 8, 16, 0, 1) X(7, 8, 4, 0, 1)
#else
#define MG_RENDER_GROUP_V(X)
#endif
#define MG_RENDER_GROUP_N(X) /* mg_step_render_encode (V + 16): the BASELINE configs' shapes, the default tile, any view at tile 8 */ \
    X(7, 8, 16, 16, 0) X(7, 8, 4, 16, 0) X(9, 8, 16, 16, 0) X(9, 8, 4, 16, 0) X(0, 8, 8, 16, 0) X(0, 8, 4, 16, 0)                       \
    X(7, 5, 16, 16, 2) X(7, 5, 4, 16, 2)
#define MG_RENDER_EXTERN(VS, TS, WPB, V, RM)                                                                               \
    extern template hipError_t launch_render_t<VS, TS, WPB, V, RM>(const MgConfig&, const MgState&, uint8_t*, uint8_t*,     \
                                                                 uint8_t*, uint8_t*, hipStream_t, const FusedStep*, RenderPick*);
#define MG_RENDER_INSTANTIATE(VS, TS, WPB, V, RM)                                                                          \
    template hipError_t launch_render_t<VS, TS, WPB, V, RM>(const MgConfig&, const MgState&, uint8_t*, uint8_t*, uint8_t*,  \
                                                          uint8_t*, hipStream_t, const FusedStep*, RenderPick*);

}  // namespace mg

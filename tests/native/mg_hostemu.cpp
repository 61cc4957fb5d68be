// mg_hostemu.cpp — TEST INFRASTRUCTURE, not product code.
//
// The lane-per-env bodies of the engine (marlgrid_amd/csrc/mg_core.h: seed / reset / step / live
// placement) compiled for the host with g++ and driven one env at a time, so that the state
// machine the HIP kernels run per lane can be stepped against the oracle in THIS container, which
// has no GPU (tests/test_core_hostemu.py).  Nothing under marlgrid_amd/ loads this library, and it
// renders no observations: the obs raster exists only as HIP kernels.
#include <string.h>

#include <vector>

#include "mg_core.h"

// the gather raster's index arithmetic (mg_gather.h), every LDS offset it forms checked against the buffers
static uint32_t g_gather_tmap_bytes = 0, g_gather_atlas_bytes = 0, g_gather_oob = 0;
#define MG_GATHER_BOUNDS(tmap_off, atlas_off)                                                               \
    do {                                                                                                    \
        if ((uint32_t)(tmap_off) + 2u > g_gather_tmap_bytes || (uint32_t)(atlas_off) + 20u > g_gather_atlas_bytes) g_gather_oob++; \
    } while (0)
#include "mg_gather.h"

// mg_encode's phases (mg_encode_core.h), every LDS offset its chunk phase forms checked against the planes
static uint32_t g_enc_plane_bytes = 0, g_enc_oob = 0;
#define MG_ENC_BOUNDS(off, bytes)                                                                           \
    do {                                                                                                    \
        if ((off) < 0 || (uint32_t)(off) + (uint32_t)(bytes) > g_enc_plane_bytes) g_enc_oob++;              \
    } while (0)
#include "mg_encode_core.h"

namespace {
struct Scratch {
    std::vector<uint64_t> rec;
    std::vector<uint32_t> head;
    std::vector<uint8_t> act, fb, oflags, ord;
    mg::StepScratch sc;
    Scratch(const MgConfig* cfg) : rec(MG_MAX_AGENTS), head(MG_MT_HEAD), act(MG_MAX_AGENTS),
                                   fb(MG_MAX_AGENTS), oflags(MG_MAX_OBJ, 0), ord(MG_MAX_AGENTS) {
        for (int i = 1; i < cfg->n_obj; i++) oflags[i] = cfg->obj[i].flags;
        sc.rec = rec.data(); sc.head = head.data(); sc.act = act.data(); sc.fb = fb.data(); sc.ord = ord.data();
        sc.obj = cfg->obj; sc.oflags = oflags.data(); sc.S = 1; sc.col = 0;
    }
};
}  // namespace

extern "C" {

int emu_mt_seed(int B, const uint32_t* keys, const int32_t* key_len, uint32_t* mt, int32_t* mt_pos, uint32_t* head) {
    for (int b = 0; b < B; b++)
        mg::mt_seed_env(keys + (size_t)b * MG_KEY_WORDS, key_len[b], mt + (size_t)b * MG_MT_N, mt_pos + b,
                        head + (size_t)b * MG_MT_HEAD);
    return 0;
}

int emu_reset(const MgConfig* cfg, const MgState* st, const MgGenProgram* prog, const uint8_t* mask) {
    Scratch s(cfg);
    for (int b = 0; b < cfg->B; b++) {
        if (mask && !mask[b]) continue;
        mg::reset_run(*cfg, *st, *prog, s.oflags.data(), b, mask != st->done, s.rec.data(), 1, 0);
    }
    return 0;
}

int emu_step(const MgConfig* cfg, const MgState* st, const void* actions, int action_bytes, float* rewards,
             const MgGenProgram* auto_reset) {
    Scratch s(cfg);
    MgGenProgram none;
    memset(&none, 0, sizeof(none));
    const MgGenProgram& prog = auto_reset ? *auto_reset : none;
    for (int b = 0; b < cfg->B; b++) {
        if (action_bytes != 1 && action_bytes != 4 && action_bytes != 8) return -100;
        // odd envs without the pre-loaded front cells (StepScratch::fb == nullptr: the obs kernel's fused step reads its
        // staged grid directly), even envs with them (mg_step's kernel): the oracle comparison covers both
        s.sc.fb = (b & 1) ? nullptr : s.fb.data();
        const mg::StepEnv e = mg::step_load(*cfg, *st, actions, action_bytes, b, s.sc);
        // stepped on a staged copy of the grid slice, as the obs kernel does (mg_render.hip): the copy goes
        // back only when step_run reports it written — and must be unchanged when it does not
        uint8_t* home = st->grid + (size_t)b * cfg->cells_stride;
        std::vector<uint8_t> staged(home, home + cfg->cells_stride);
        // ... and every third env with the write-back left to the caller, as the obs kernel does it for its whole batch
        s.sc.defer_writeback = (b % 3) == 2;
        const mg::StepOut out = mg::step_run(*cfg, *st, prog, auto_reset != nullptr, rewards, b, e, s.sc, staged.data());
        if (s.sc.defer_writeback) {
            for (int k = 0; k < cfg->n_agents; k++) st->agents[(size_t)b * cfg->n_agents + k] = s.rec[k];
            for (int j = 0; j < MG_MT_HEAD; j++) st->mt_head[(size_t)b * MG_MT_HEAD + j] = s.head[(j + out.head_k) & (MG_MT_HEAD - 1)];
        }
        const bool wrote = out.wrote;
        if (wrote) memcpy(home, staged.data(), cfg->cells_stride);
        else if (memcmp(home, staged.data(), cfg->cells_stride) != 0) return -101;
    }
    return 0;
}

// The obs kernel's fused step as a wave runs it (mg_render_kernel.h): batches of up to 8 staged envs in S = 8 columns,
// step_begin on the env's lane, the agents resolved by one lane per (agent, env) — step_par_publish / _resolve / _commit,
// here lane after lane with the phases in the kernel's order —, the sequential loop only for the envs that asked for it,
// step_end.  *n_serial counts those envs.  More than 8 agents: the sequential step, as in the kernel.
int emu_step_par(const MgConfig* cfg, const MgState* st, const void* actions, int action_bytes, float* rewards,
                 const MgGenProgram* auto_reset, int64_t* n_serial) {
    if (cfg->n_agents > 8) return emu_step(cfg, st, actions, action_bytes, rewards, auto_reset);
    if (action_bytes != 1 && action_bytes != 4 && action_bytes != 8) return -100;
    const int n = cfg->n_agents, stride = cfg->cells_stride;
    MgGenProgram none;
    memset(&none, 0, sizeof(none));
    const MgGenProgram& prog = auto_reset ? *auto_reset : none;
    std::vector<uint64_t> rec(n * 8), rec_out(n * 8);
    std::vector<uint32_t> head(MG_MT_HEAD * 8);
    std::vector<uint8_t> act(n * 8), pflag(n * 8), ordp(n * 8), oflags(MG_MAX_OBJ, 0), grids((size_t)8 * stride);
    std::vector<int32_t> psc(8);
    for (int i = 1; i < cfg->n_obj; i++) oflags[i] = cfg->obj[i].flags;
    for (int b0 = 0; b0 < cfg->B; b0 += 8) {
        const int kb = cfg->B - b0 < 8 ? cfg->B - b0 : 8;
        mg::StepScratch sc;
        sc.rec = rec.data(); sc.head = head.data(); sc.act = act.data(); sc.fb = nullptr;
        sc.obj = cfg->obj; sc.oflags = oflags.data(); sc.S = 8; sc.col = 0;
        sc.pflag = pflag.data(); sc.ordp = ordp.data(); sc.psc = psc.data(); sc.rec_out = rec_out.data();
        sc.defer_writeback = true;
        memset(pflag.data(), 0xEE, pflag.size());
        memset(ordp.data(), 0xEE, ordp.size());
        mg::StepCtx ctx[8];
        for (int j = 0; j < kb; j++) {
            sc.col = j;
            memcpy(grids.data() + (size_t)j * stride, st->grid + (size_t)(b0 + j) * stride, stride);
            const mg::StepEnv e = mg::step_load(*cfg, *st, actions, action_bytes, b0 + j, sc);
            ctx[j] = mg::step_begin(*cfg, *st, b0 + j, e, sc, grids.data() + (size_t)j * stride);
            mg::step_par_publish(*cfg, sc, ctx[j]);
        }
        mg::ParLane P[64];
        bool serial[64];
        for (int lane = 0; lane < 64; lane++) P[lane] = mg::step_par_resolve(*cfg, sc, grids.data(), kb, lane);
        for (int lane = 0; lane < 64; lane++) serial[lane] = mg::step_par_commit(*cfg, *st, rewards, b0, sc, P[lane], lane);
        for (int lane = 0; lane < 64; lane++)
            if (P[lane].live && !serial[lane]) rec[lane] = rec_out[lane];        // (on the GPU rec_out IS rec: the lanes run in lockstep)
        for (int j = 0; j < kb; j++) {
            const int b = b0 + j;
            sc.col = j;
            uint8_t* g = grids.data() + (size_t)j * stride;
            for (int k = 1; k < n; k++)
                if (serial[k * 8 + j] != serial[j]) return -102;                      // every lane of an env gives the same answer
            if (serial[j]) { mg::step_agents(*cfg, *st, rewards, b, sc, g, ctx[j]); if (n_serial) (*n_serial)++; }
            const mg::StepOut out = mg::step_end(*cfg, *st, prog, auto_reset != nullptr, b, sc, g, ctx[j]);
            for (int k = 0; k < n; k++) st->agents[(size_t)b * n + k] = rec[k * 8 + j];
            for (int i = 0; i < MG_MT_HEAD; i++) st->mt_head[(size_t)b * MG_MT_HEAD + i] = head[((i + out.head_k) & (MG_MT_HEAD - 1)) * 8 + j];
            uint8_t* home = st->grid + (size_t)b * stride;
            if (out.wrote) memcpy(home, g, stride);
            else if (memcmp(home, g, stride) != 0) return -101;
        }
    }
    return 0;
}

int emu_place(const MgConfig* cfg, const MgState* st, int what, int x0, int y0, int x1, int y1, int max_tries,
              const int32_t* fixed_pos, const uint8_t* mask, const uint8_t* reject, int32_t* out_pos, uint8_t* out_ok) {
    Scratch s(cfg);
    for (int b = 0; b < cfg->B; b++) {
        if (mask && !mask[b]) continue;
        mg::place_run(*cfg, *st, s.oflags.data(), b, what, x0, y0, x1, y1, max_tries, fixed_pos, reject, out_pos, out_ok,
                      s.rec.data(), 1, 0);
    }
    return 0;
}

}  // extern "C"

namespace {
// The obs kernel's gather raster for one GROUP of envs as one wave runs it: the padded atlas built dword by dword as
// the kernel's prologue does (pad_source / pad_cut), then gather_group lane by lane (the lanes of the raster do not
// talk to each other).  Returns the number of out-of-range LDS offsets formed (0 = none), -1 for bad arguments.
template <int VS, int TS>
int gather_emu(int n_vt, int n_dyn, const uint8_t* atlas_raw, const uint16_t* tmap, int tmap_entries, uint8_t* dst, uint32_t stream_bytes) {
    typedef mg::GatherGeom<VS, TS> Gm;
    // virtual tiles [0, n_vt): the atlas, padded dword by dword as the kernel's prologue does; [n_vt, n_vt + n_dyn): a
    // wave's own recoloured tiles, in the same padded layout somewhere else in LDS (here: behind a gap of garbage)
    const int rows = n_vt * TS, raw16 = (rows * Gm::SEG + 15) / 16 * 16;
    std::vector<uint8_t> raw(raw16, 0);
    memcpy(raw.data(), atlas_raw, (size_t)rows * Gm::SEG);
    constexpr int ROW_W = Gm::RS / 4;
    const int npd = rows * ROW_W + Gm::TAIL / 4, gap = 37 * 4, dyn_w = n_dyn * TS * ROW_W + (n_dyn ? Gm::TAIL / 4 : 0);
    std::vector<uint32_t> padded(npd + gap / 4 + dyn_w, 0xA5A5A5A5u);
    for (int d = 0; d < npd; d++) {
        uint32_t cut, keep;
        const int a = mg::pad_source<Gm::SEG, Gm::FRONT / 4, ROW_W>(d, rows, raw16, cut, keep);
        uint32_t lo = 0, hi = 0;
        if (a >= 0) { memcpy(&lo, raw.data() + a, 4); memcpy(&hi, raw.data() + a + 4, 4); }
        padded[d] = a >= 0 ? mg::pad_cut(lo, hi, cut, keep) : 0u;
    }
    uint8_t* dynp = reinterpret_cast<uint8_t*>(padded.data() + npd) + gap;
    if (n_dyn) memset(dynp, 0, (size_t)dyn_w * 4);
    for (int t = 0; t < n_dyn; t++)
        for (int r = 0; r < TS; r++)
            memcpy(dynp + (size_t)t * Gm::TILE + r * Gm::RS + Gm::FRONT, atlas_raw + ((size_t)(n_vt + t) * TS + r) * Gm::SEG, Gm::SEG);
    const mg::GatherDyn dyn = {(uint32_t)n_vt, (uint32_t)(npd * 4 + gap) - (uint32_t)n_vt * Gm::TILE};
    g_gather_tmap_bytes = (uint32_t)tmap_entries * 2u;
    g_gather_atlas_bytes = (uint32_t)padded.size() * 4u;
    g_gather_oob = 0;
    for (int lane = 0; lane < 64; lane++) {
        if (n_dyn) mg::gather_group<VS, TS, true>(lane, reinterpret_cast<const uint8_t*>(tmap), reinterpret_cast<const uint8_t*>(padded.data()), dst, stream_bytes, dyn);
        else mg::gather_group<VS, TS>(lane, reinterpret_cast<const uint8_t*>(tmap), reinterpret_cast<const uint8_t*>(padded.data()), dst, stream_bytes);
    }
    return (int)g_gather_oob;
}
}  // namespace

extern "C" {

// geometry of (vs, ts) as the kernel sees it: out[0..7] = SEG, RS, PC, PR, C, NT, LPT, kConstBand
int emu_gather(int vs, int ts, int n_vt, int n_dyn, const uint8_t* atlas_raw, const uint16_t* tmap, int tmap_entries, uint8_t* dst,
               uint32_t stream_bytes, int32_t* geom) {
#define MG_EMU_GATHER(VS, TS)                                                                                           \
    if (vs == VS && ts == TS) {                                                                                         \
        typedef mg::GatherGeom<VS, TS> Gm;                                                                              \
        if (geom) { const int32_t g[8] = {Gm::SEG, Gm::RS, Gm::PC, Gm::PR, Gm::C, Gm::NT, Gm::LPT, Gm::kConstBand};    \
                    memcpy(geom, g, sizeof(g)); }                                                                       \
        return gather_emu<VS, TS>(n_vt, n_dyn, atlas_raw, tmap, tmap_entries, dst, stream_bytes);                              \
    }
    MG_EMU_GATHER(7, 5) MG_EMU_GATHER(7, 6) MG_EMU_GATHER(7, 7) MG_EMU_GATHER(7, 9) MG_EMU_GATHER(7, 10) MG_EMU_GATHER(7, 11)
    MG_EMU_GATHER(7, 12) MG_EMU_GATHER(5, 5) MG_EMU_GATHER(9, 6) MG_EMU_GATHER(3, 5) MG_EMU_GATHER(6, 5) MG_EMU_GATHER(4, 6) MG_EMU_GATHER(9, 5) MG_EMU_GATHER(4, 5) MG_EMU_GATHER(8, 5) MG_EMU_GATHER(11, 5) MG_EMU_GATHER(13, 5) MG_EMU_GATHER(15, 5)
#undef MG_EMU_GATHER
    return -1;
}

// mg_encode as its kernel runs it: piece by piece, phase by phase (a barrier between phases), thread by thread; the LDS
// of a workgroup is a buffer of exactly the size the launcher asks for, filled with garbage first.  pc: 0 = the launcher's
// choice, or 8192 / 4096 / 1024.  Returns the number of out-of-range LDS offsets formed (0 = none), -1 for bad arguments.
static int g_enc_force_runs = 0;      // 1 / 2: the store phase of encode_runs as large / small batches take it (0: as the launcher decides)
void emu_encode_force_runs(int v) { g_enc_force_runs = v; }
int emu_encode(const MgConfig* cfg, const MgState* st, const uint8_t* vis, uint8_t* out, int pc) {
    if (pc != 0 && pc != 8192 && pc != 4096 && pc != 1024) return -1;
    int PC = pc;
    mg::EncodeLaunch lc = mg::encode_launch(*cfg, out, PC);
    if (lc.runs && g_enc_force_runs) lc.runs = g_enc_force_runs;
    const int T = PC / 16;
    const size_t lds = mg::kEncTab + (size_t)lc.nraw * (lc.two ? 2 : 1) + (lc.runs ? 3 * (size_t)PC : 0);
    g_enc_plane_bytes = (uint32_t)lc.nraw;
    g_enc_oob = 0;
    const long long pieces = (lc.total + PC - 1) / PC;
    std::vector<uint8_t> smem(lds + 16);
    uint8_t* sm = smem.data() + ((16 - (reinterpret_cast<uintptr_t>(smem.data()) & 15)) & 15);
    for (long long b = 0; b < pieces; b++) {
        memset(sm, 0xA5, lds);
        const mg::EncodePiece P = mg::encode_piece(*cfg, lc, b, PC);
        if ((size_t)P.nd * 4 > (size_t)lc.nraw) return -2;
        for (int tid = 0; tid < T; tid++) mg::encode_stage(*cfg, *st, lc, P, sm, tid, T);
        for (int tid = 0; tid < T; tid++) mg::encode_agents(*cfg, *st, lc, P, sm, tid, T);
        int q_first = 0;        // (as the kernel: by runs of 16 cells where it can, the chunk form for the rest)
        if (lc.runs && !vis) {
            for (int tid = 0; tid < T; tid++) q_first = mg::encode_runs(*cfg, lc, P, sm, tid, T);
            for (int tid = 0; tid < T; tid++) mg::encode_runs_store(lc, P, out, sm, tid, T);
        }
        for (int tid = 0; tid < T; tid++) mg::encode_chunks(*cfg, lc, P, vis, out, sm, tid, T, PC, q_first);
    }
    return (int)g_enc_oob;
}

// The fused step's encode (encode_batch_mark / encode_batch_chunks) as a wave of the obs kernel runs it: the wave's batch of
// kb <= 8 envs staged in "LDS" — the grids in HBM layout, the records env by env —, 64 lanes, the table as the kernel's
// prologue builds it.  per_wave: envs a wave walks (its batches follow each other).  slack: bytes of the staging buffer
// behind the batch's grids that may be read (never used): what follows the grids in the wave's scratch.
int emu_encode_batch(const MgConfig* cfg, const MgState* st, uint8_t* out, int per_wave, int rec_stride) {
    const int n = cfg->n_agents, cells = cfg->W * cfg->H, stride = cfg->cells_stride;
    if (cfg->n_obj + 4 * n > 256 || rec_stride < n) return -1;
    std::vector<uint32_t> tab(256, 0xDEADBEEFu);
    for (int o = 0; o < cfg->n_obj; o++)
        tab[o] = o ? ((uint32_t)cfg->obj[o].type_idx | ((uint32_t)cfg->obj[o].color_idx << 8) | ((uint32_t)cfg->obj[o].state << 16)) : 0u;
    for (int code = 0; code < 4 * n; code++)
        tab[cfg->n_obj + code] = (uint32_t)cfg->agent_type_idx | ((uint32_t)cfg->agent_color_idx[code >> 2] << 8) | ((uint32_t)(code & 3) << 16);
    const uint32_t m_cells = (uint32_t)((0x100000000ull + (uint32_t)cells - 1) / (uint32_t)cells);
    const uint32_t m_n = (uint32_t)((0x100000000ull + (uint32_t)n - 1) / (uint32_t)n);
    const int K = 8, slack = 16;
    std::vector<uint8_t> raw((size_t)K * stride + slack + 16);
    uint8_t* rw = raw.data() + ((16 - (reinterpret_cast<uintptr_t>(raw.data()) & 15)) & 15);
    std::vector<uint64_t> recs((size_t)K * rec_stride);
    g_enc_plane_bytes = (uint32_t)(K * stride + slack);
    g_enc_oob = 0;
    for (int e0 = 0; e0 < cfg->B; e0 += per_wave)
        for (int eb = e0; eb < e0 + per_wave && eb < cfg->B; eb += K) {
            int kb = e0 + per_wave - eb;
            if (kb > K) kb = K;
            if (kb > cfg->B - eb) kb = cfg->B - eb;
            memset(rw, 0xA5, (size_t)K * stride + slack);
            memcpy(rw, st->grid + (size_t)eb * stride, (size_t)kb * stride);
            for (int j = 0; j < kb; j++)
                for (int k = 0; k < n; k++) recs[(size_t)j * rec_stride + k] = st->agents[(size_t)(eb + j) * n + k];
            for (int lane = 0; lane < 64; lane++) mg::encode_batch_mark(*cfg, rw, recs.data(), rec_stride, kb, m_n, lane, 64);
            for (int lane = 0; lane < 64; lane++)
                mg::encode_batch_chunks(*cfg, rw, tab.data(), out, (long long)eb * cells, kb, cells, m_cells, lane, 64);
            // the marks taken out again: the staged grids are what they were (the views that follow read them)
            for (int lane = 0; lane < 64; lane++) mg::encode_batch_mark(*cfg, rw, recs.data(), rec_stride, kb, m_n, lane, 64, true);
            if (memcmp(rw, st->grid + (size_t)eb * stride, (size_t)kb * stride) != 0) return -3;
        }
    return (int)g_enc_oob;
}

int emu_sizeof(int which) {
    switch (which) {
    case 0: return (int)sizeof(MgConfig);
    case 1: return (int)sizeof(MgState);
    case 2: return (int)sizeof(MgGenProgram);
    case 3: return (int)sizeof(MgObjDesc);
    default: return -1;
    }
}

}  // extern "C"

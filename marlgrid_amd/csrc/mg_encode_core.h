// mg_encode_core.h — the phases of mg_encode.hip's kernel as plain inline functions (thread `tid` of a workgroup of T; LDS as a
// pointer), so that the same text compiles with g++ for tests/native, which runs a piece phase by phase, thread by thread,
// against a cell-by-cell encode on the host before any GPU sees it.
//
// mg_encode — batched MultiGrid.encode (marlgrid/base.py:196-214): per cell the (type_idx, colour_idx, state) triple
// of the cell's *top object* (WorldObj.encode, objects.py:90-99); agents stacked on another object are not reflected, an
// agent that is the cell object encodes as (13, colour, dir).  out: uint8 [B][W][H][3].
//
// Roofline: HBM.  Algorithmic bytes per env = cells_stride + 8 n (in) + 3 W H (out): 939 B at 15x15 with three agents.
//
// The output of a batch is ONE flat byte stream (3 bytes per cell, cells of consecutive envs back to back — an env's
// 675 bytes are not a multiple of anything), so the kernel is written from the output's side: a PIECE is PC consecutive
// cells of the flat [B * W*H] cell space (PC a multiple of 16: a piece's 3 PC bytes are whole aligned 16-byte chunks) and
// belongs to one workgroup of PC / 16 threads:
//   A. the piece's slice of `grid` — one contiguous run of HBM, the 15 padding bytes between envs included — goes to LDS
//      as it is (aligned dwords, every load in flight before the first wait), the (type, colour, state) table of the
//      object kinds (one dword per kind) next to it;
//   B. one lane per (env, agent) of the envs the piece touches: an agent that is the FIRST of its cell (lowest arrival
//      rank: `obj.agents[0]` / the cell object, base.py:547-552) marks the cell — in the grid bytes themselves where object
//      ids and agent codes share a byte (n_obj + 4 n <= 256), in a second byte plane otherwise;
//   C. a lane per 16-byte chunk of output: the six cells the chunk spans (the second half of them possibly in the next
//      env: two aligned LDS windows merged by a 64-bit shift), six table look-ups, the 18 bytes packed with v_perm_b32
//      and cut at the chunk's phase with v_alignbyte, one global_store_dwordx4.
// Nothing per cell but a table look-up; no division per cell (one per chunk); every store 16 bytes and aligned.
// (Round 2's kernel — a lane per cell, a 64-bit divide, a 32-byte descriptor gather and n record loads per empty cell,
// three byte stores — ran at 0.11 of the HBM roofline: profiles/r02/README.md.)
#pragma once
#include "mg_core.h"

namespace mg {

struct EncodeLaunch {
    long long total;          // B * W * H cells
    int cells, nraw;          // W * H; bytes of the raw grid plane in LDS (a multiple of 16)
    uint32_t m_cells, m_n;    // divide-by-multiply constants (32-bit mul_hi): ceil(2^32 / cells), ceil(2^32 / n)
    int two, aligned;         // agent marks in their own byte plane; `out` is 16-byte aligned
    int runs;                 // phase C by runs of 16 cells (encode_runs): one byte plane, aligned `out`, W * H >= 16; 2: ... and its
                              // waves stream their own runs out (small batches)
};

MG_HD uint32_t enc_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
MG_HD uint32_t enc_align(uint32_t hi, uint32_t lo, uint32_t sh) {     // bytes [sh, sh + 4) of lo | hi << 32, sh = 0..3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh));
#endif
}
MG_HD uint32_t enc_perm(uint32_t hi, uint32_t lo, uint32_t sel) {     // v_perm_b32: selector byte 0..3 = lo's bytes, 4..7 = hi's, 12 = 0x00
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFFu;
        const uint32_t b = s < 4 ? (lo >> (8 * s)) & 0xFFu : s < 8 ? (hi >> (8 * (s - 4))) & 0xFFu : 0u;
        out |= b << (8 * i);
    }
    return out;
#endif
}
MG_HD uint32_t enc_div(uint32_t x, uint32_t d, uint32_t m) {     // x / d, m = ceil(2^32 / d): exact for every 32-bit x
    if (d == 1) return x;                       // (m would be 2^32)
    uint32_t q = enc_mulhi(x, m);
    q -= (q * d > x) ? 1u : 0u;
    return q;
}

// the piece of workgroup `block`: where it starts, what it spans
struct EncodePiece {
    long long g0, b0;         // first cell (flat), its env
    int len, c0, c0a;         // cells; the first cell within its env, rounded down to a dword
    int el, nd;               // the last cell lies in env b0 + el; dwords of the slice
};
MG_HD EncodePiece encode_piece(const MgConfig& cfg, const EncodeLaunch& lc, long long block, int PC) {
    EncodePiece P;
    P.g0 = block * PC;
    P.len = (int)(lc.total - P.g0 < (long long)PC ? lc.total - P.g0 : (long long)PC);
    // (uniform: one division per workgroup — by multiply-high while the batch has fewer than 2^32 cells: as a 64-bit division it
    // was ~200 dependent scalar instructions in front of the workgroup's first load)
    P.b0 = lc.total <= 0xFFFFFFFFll ? (long long)enc_div((uint32_t)P.g0, (uint32_t)lc.cells, lc.m_cells) : P.g0 / lc.cells;
    P.c0 = (int)(P.g0 - P.b0 * lc.cells);
    P.c0a = P.c0 & ~3;
    P.el = (int)enc_div((uint32_t)(P.c0 + P.len - 1), (uint32_t)lc.cells, lc.m_cells);
    const int cl = P.c0 + P.len - 1 - P.el * lc.cells;
    P.nd = (P.el * cfg.cells_stride + cl + 1 - P.c0a + 3) >> 2;
    return P;
}
constexpr int kEncRecCap = 512;    // agent records of a piece's envs staged in LDS (more — tiny grids with many agents —: read in place)
constexpr int kEncTab = 2048 + kEncRecCap * 8;     // bytes of the two look-up tables and the staged records in front of the planes

// A. the slice of `grid` and the tables -> LDS.  (The kernel issues its first 8 dwords per thread before anything else.)
MG_HD void encode_stage(const MgConfig& cfg, const MgState& st, const EncodeLaunch& lc, const EncodePiece& P, uint8_t* smem, int tid, int T) {
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);                     // [256] object id (or agent code) -> triple
    uint32_t* tab2 = tab + 256;                                             // [256] two planes: agent code -> triple (0: none)
    uint64_t* lrec = reinterpret_cast<uint64_t*>(smem + 2048);              // [kEncRecCap] the records of the piece's envs
    uint8_t* raw = smem + kEncTab;                                          // the piece's slice of `grid`, HBM layout
    uint8_t* ag = raw + lc.nraw;                                            // (two planes) agent marks, same layout
    const int n = cfg.n_agents;
    const uint32_t* gsrc = reinterpret_cast<const uint32_t*>(st.grid + (size_t)P.b0 * cfg.cells_stride + P.c0a);
    constexpr int R = 8;
    uint32_t v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = (tid + r * T < P.nd) ? gsrc[tid + r * T] : 0u;
    // the records of the envs the piece touches ride the same round trip (one contiguous run; the agents' phase then reads
    // LDS only: as global loads behind the barrier they were a second dependent round trip of every workgroup)
    const int nrec = (P.el + 1) * n;
    const uint64_t* rsrc = st.agents + (size_t)P.b0 * n;
    uint64_t rv[2] = {0ull, 0ull};
    if (nrec <= kEncRecCap) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (tid + r * T < nrec) rv[r] = rsrc[tid + r * T];
    }
    // the tables: at most four entries per thread (T >= 64), every load requested before the first is used (as a loop of
    // load-then-store trips the one-wave workgroups of small batches made four dependent round trips of it)
    uint32_t te[4], ta[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int o = tid + q * T;
        uint32_t e = 0, a = 0;
        if (o > 0 && o < cfg.n_obj)                                         // type, colour, state: the descriptor's first three bytes
            e = *reinterpret_cast<const uint32_t*>(cfg.obj + o) & 0xFFFFFFu;
        // agent codes: agent k facing d -> (agent_type_idx, colour of k, d)
        const int code = lc.two ? o - 1 : o - cfg.n_obj;
        if (o < 256 && code >= 0 && code < 4 * n) a = (uint32_t)cfg.agent_type_idx | ((uint32_t)cfg.agent_color_idx[code >> 2] << 8) | ((uint32_t)(code & 3) << 16);
        te[q] = e; ta[q] = a;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int o = tid + q * T;
        if (o < 256) {
            if (lc.two) { tab[o] = te[q]; tab2[o] = ta[q]; }
            else tab[o] = (o < cfg.n_obj) ? te[q] : ta[q];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        if (tid + r * T < P.nd) reinterpret_cast<uint32_t*>(raw)[tid + r * T] = v[r];
    for (int i = tid + R * T; i < P.nd; i += T) reinterpret_cast<uint32_t*>(raw)[i] = gsrc[i];
    if (nrec <= kEncRecCap) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (tid + r * T < nrec) lrec[tid + r * T] = rv[r];
        for (int i = tid + 2 * T; i < nrec; i += T) lrec[i] = rsrc[i];
    }
    if (lc.two)
        for (int i = tid; i < lc.nraw / 4; i += T) reinterpret_cast<uint32_t*>(ag)[i] = 0u;
}

// B. the first agent of every occupied cell
MG_HD void encode_agents(const MgConfig& cfg, const MgState& st, const EncodeLaunch& lc, const EncodePiece& P, uint8_t* smem, int tid, int T) {
    uint8_t* raw = smem + kEncTab;
    uint8_t* ag = raw + lc.nraw;
    const uint64_t* lrec = reinterpret_cast<const uint64_t*>(smem + 2048);
    const int n = cfg.n_agents, cells = lc.cells, stride = cfg.cells_stride;
    const int items = (P.el + 1) * n;
    const bool staged = items <= kEncRecCap;
    for (int it = tid; it < items; it += T) {
        const uint32_t e = enc_div((uint32_t)it, (uint32_t)n, lc.m_n);
        const int k = it - (int)e * n;
        const uint64_t* recs = staged ? lrec + (size_t)e * n : st.agents + (size_t)(P.b0 + e) * n;
        const uint64_t r = recs[k];
        if (!(rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED)) continue;
        const int c = (int)rec_byte(r, MG_AG_X) * cfg.H + (int)rec_byte(r, MG_AG_Y);
        const int f = (int)e * cells + c - P.c0;                            // the cell, piece-relative
        if (f < 0 || f >= P.len) continue;
        const uint32_t xy = rec_xy(r), rank = rec_byte(r, MG_AG_RANK);
        bool first = true;
        for (int j = 0; j < n; j++) {
            const uint64_t rj = recs[j];
            if ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy && rec_byte(rj, MG_AG_RANK) < rank) first = false;
        }
        if (!first) continue;
        const int addr = (int)e * stride + c - P.c0a;
        const uint32_t code = 4u * (uint32_t)k + rec_byte(r, MG_AG_DIR);
        if (lc.two) ag[addr] = (uint8_t)(1u + code);
        else if (raw[addr] == 0) raw[addr] = (uint8_t)((uint32_t)cfg.n_obj + code);     // (only an empty cell's object is the agent)
    }
}

// C. 16-byte chunks of output
#if !defined(MG_ENC_BOUNDS)
#define MG_ENC_BOUNDS(off, bytes) do {} while (0)
#endif
MG_HD void encode_chunks(const MgConfig& cfg, const EncodeLaunch& lc, const EncodePiece& P, const uint8_t* vis, uint8_t* out,
                         const uint8_t* smem, int tid, int T, int PC, int q_first = 0) {
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t* tab2 = tab + 256;
    const uint8_t* raw = smem + kEncTab;
    const uint8_t* ag = raw + lc.nraw;
    const int cells = lc.cells, stride = cfg.cells_stride;
    uint8_t* dst = out + (size_t)P.g0 * 3;
    const int nbytes = P.len * 3;
    const int NQ = 3 * PC / 16;
    for (int q = q_first + tid; q < NQ; q += T) {
        if (16 * q >= nbytes) break;
        const uint32_t qd = ((uint32_t)q * 21846u) >> 16;                   // q / 3 (q < 32768)
        const uint32_t p = (uint32_t)q - 3u * qd;                           // the chunk's first byte is byte p of its first cell
        const int f = 16 * (int)qd + (p == 0 ? 0 : p == 1 ? 5 : 10);        // ... which is cell f of the piece
        const uint32_t cc = (uint32_t)(P.c0 + f);
        const uint32_t e = enc_div(cc, (uint32_t)cells, lc.m_cells);
        const int c = (int)cc - (int)e * cells;
        const int nA = cells - c;                                           // cells left in this env, this one included
        const int addrA = (int)e * stride + c - P.c0a;
        // six consecutive cells: a window of the env's bytes, continued — where the env ends inside it — by the next env's first
        const bool needB = nA < 6 && f + nA < P.len;
        const int addrB = needB ? ((int)e + 1) * stride - P.c0a : 0;        // (4-byte aligned: stride is a multiple of 16)
        MG_ENC_BOUNDS(addrA & ~3, 12);
        MG_ENC_BOUNDS(addrB, 8);
        uint64_t X, Y = 0;
        {
            const uint32_t sh = (uint32_t)addrA & 3u, s8 = 8u * (uint32_t)(nA < 7 ? nA : 7);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(raw + (addrA & ~3));
            const uint32_t* wb = reinterpret_cast<const uint32_t*>(raw + addrB);
            const uint64_t A = (uint64_t)enc_align(w[1], w[0], sh) | ((uint64_t)enc_align(w[2], w[1], sh) << 32);
            const uint64_t Bv = (uint64_t)wb[0] | ((uint64_t)wb[1] << 32);
            X = needB ? ((A & ~(~0ull << s8)) | (Bv << s8)) : A;
            if (lc.two) {
                const uint32_t* u = reinterpret_cast<const uint32_t*>(ag + (addrA & ~3));
                const uint32_t* ub = reinterpret_cast<const uint32_t*>(ag + addrB);
                const uint64_t A2 = (uint64_t)enc_align(u[1], u[0], sh) | ((uint64_t)enc_align(u[2], u[1], sh) << 32);
                const uint64_t B2 = (uint64_t)ub[0] | ((uint64_t)ub[1] << 32);
                Y = needB ? ((A2 & ~(~0ull << s8)) | (B2 << s8)) : A2;
            }
        }
        uint32_t t[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const uint32_t b = (uint32_t)(X >> (8 * i)) & 0xFFu;
            t[i] = tab[b];
            if (lc.two && b == 0) t[i] = tab2[(uint32_t)(Y >> (8 * i)) & 0xFFu];
        }
        if (vis) {      // vis_mask (base.py:205-206): cells that are not visible encode as (0, 0, 0); flat [B][W][H]
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (f + i < P.len && !vis[(size_t)P.g0 + f + i]) t[i] = 0;
        }
        // 18 bytes = 6 x 3, as five dwords; the chunk is bytes p .. p + 15 of them
        const uint32_t d0 = enc_perm(t[1], t[0], 0x04020100u);      // t0.0 t0.1 t0.2 t1.0
        const uint32_t d1 = enc_perm(t[2], t[1], 0x05040201u);      // t1.1 t1.2 t2.0 t2.1
        const uint32_t d2 = enc_perm(t[3], t[2], 0x06050402u);      // t2.2 t3.0 t3.1 t3.2
        const uint32_t d3 = enc_perm(t[5], t[4], 0x04020100u);      // t4.0 t4.1 t4.2 t5.0
        const uint32_t d4 = enc_perm(0u, t[5], 0x0C0C0201u);        // t5.1 t5.2 0 0
        uint32_t w4[4] = {enc_align(d1, d0, p), enc_align(d2, d1, p), enc_align(d3, d2, p), enc_align(d4, d3, p)};
        if (lc.aligned && 16 * q + 16 <= nbytes) {
            typedef struct { uint32_t v[4]; } __attribute__((aligned(16))) q16;
            q16 o4 = {{w4[0], w4[1], w4[2], w4[3]}};
            *reinterpret_cast<q16*>(dst + 16 * q) = o4;
        } else {        // the batch's last < 16 bytes, or a caller's `out` that is not 16-byte aligned
            for (int i = 0; i < 16 && 16 * q + i < nbytes; i++) dst[16 * q + i] = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
        }
    }
}

// C'. The same output by RUNS: a lane per 16 consecutive cells = 48 output bytes = three whole aligned chunks (a piece starts at
// a multiple of 16 cells), where object ids and agent marks share the byte plane, nothing is masked (no vis_mask) and a run
// touches at most two envs (W * H >= 16): ONE division, ONE window of 16 grid bytes (five aligned dwords cut with v_alignbyte; a
// second one, the next env's first bytes, merged in byte-wise with v_bfi where the env ends inside the run), sixteen table
// look-ups, twelve v_perm_b32, three 16-byte LDS stores — 2.5 instructions per output byte where the chunk form (one division, one window
// of 6 cells, 6 look-ups, 9 permutes per 16 bytes) spends 4.4: the kernel is bound by its instructions, not by its bytes
// (measurement build: everything but the stores 8.4 of 10.5 us at 32 768 envs, 43.0 of 44.4 at 262 144; the stores alone 5.8
// / 22.6).  Returns the first chunk it did NOT write: encode_chunks finishes the piece from there (the batch's last cells).
// (a lane's 48 bytes go to the piece's OUTPUT IMAGE in LDS — `stg`, 48 bytes per run behind the planes —, and encode_runs_store
// streams that image out with a lane per chunk, consecutive lanes consecutive chunks: stored straight from the run's lane, three
// 16-byte stores 48 bytes apart per lane, every wave-store touched 24 cache lines a third each — slower than the chunk form)
MG_HD int encode_runs(const MgConfig& cfg, const EncodeLaunch& lc, const EncodePiece& P, uint8_t* smem, int tid, int T) {
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(smem);
    const uint8_t* raw = smem + kEncTab;
    uint8_t* stg = smem + kEncTab + lc.nraw;
    const int cells = lc.cells, stride = cfg.cells_stride;
    const int nrun = P.len >> 4;
    for (int r = tid; r < nrun; r += T) {
        const int f = 16 * r;
        const uint32_t cc = (uint32_t)(P.c0 + f);
        const uint32_t e = enc_div(cc, (uint32_t)cells, lc.m_cells);
        const int c = (int)cc - (int)e * cells;
        const int nA = cells - c;                                           // cells left in this env, this one included
        const int addrA = (int)e * stride + c - P.c0a;
        MG_ENC_BOUNDS(addrA & ~3, 20);
        uint32_t X[4];
        {
            const uint32_t sh = (uint32_t)addrA & 3u;
            const uint32_t* w = reinterpret_cast<const uint32_t*>(raw + (addrA & ~3));
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            X[0] = enc_align(w1, w0, sh); X[1] = enc_align(w2, w1, sh); X[2] = enc_align(w3, w2, sh); X[3] = enc_align(w4, w3, sh);
        }
        if (nA < 16 && f + nA < P.len) {     // the env ends inside the run: cells nA .. 15 are the next env's first
            const int addrB = ((int)e + 1) * stride - P.c0a - nA;           // (its cell j - nA at byte j of this window)
            MG_ENC_BOUNDS(addrB & ~3, 20);
            const uint32_t sh = (uint32_t)addrB & 3u;
            const uint32_t* w = reinterpret_cast<const uint32_t*>(raw + (addrB & ~3));
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            const uint32_t Y[4] = {enc_align(w1, w0, sh), enc_align(w2, w1, sh), enc_align(w3, w2, sh), enc_align(w4, w3, sh)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = nA - 4 * i;                                   // bytes of dword i that are still this env's
                const uint32_t mA = t >= 4 ? 0xFFFFFFFFu : t <= 0 ? 0u : (1u << (8 * t)) - 1u;
                X[i] = (X[i] & mA) | (Y[i] & ~mA);
            }
        }
        uint32_t t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) t[i] = tab[(X[i >> 2] >> (8 * (i & 3))) & 0xFFu];
        typedef struct { uint32_t v[4]; } __attribute__((aligned(16))) q16;
        q16 o[3];
#pragma unroll
        for (int k = 0; k < 4; k++) {        // four cells -> three dwords
            const uint32_t a = t[4 * k], b = t[4 * k + 1], cq = t[4 * k + 2], d = t[4 * k + 3];
            const uint32_t d0 = enc_perm(b, a, 0x04020100u);        // a.0 a.1 a.2 b.0
            const uint32_t d1 = enc_perm(cq, b, 0x05040201u);       // b.1 b.2 c.0 c.1
            const uint32_t d2 = enc_perm(d, cq, 0x06050402u);       // c.2 d.0 d.1 d.2
            const int j = 3 * k;
            o[j >> 2].v[j & 3] = d0; o[(j + 1) >> 2].v[(j + 1) & 3] = d1; o[(j + 2) >> 2].v[(j + 2) & 3] = d2;
        }
        q16* d48 = reinterpret_cast<q16*>(stg + 48 * (size_t)r);
        d48[0] = o[0]; d48[1] = o[1]; d48[2] = o[2];
    }
    return 3 * nrun;
}
MG_HD void encode_runs_store(const EncodeLaunch& lc, const EncodePiece& P, uint8_t* out, const uint8_t* smem, int tid, int T) {
    typedef struct { uint32_t v[4]; } __attribute__((aligned(16))) q16;
    const q16* stg = reinterpret_cast<const q16*>(smem + kEncTab + lc.nraw);
    q16* dst = reinterpret_cast<q16*>(out + (size_t)P.g0 * 3);
    const int nq = 3 * (P.len >> 4);
    if (lc.runs == 2) {
        // small batches: a wave streams out the 192 chunks of ITS 64 runs (run r is thread r's) — no barrier of the workgroup
        // between the two phases (-0.2 us of 9 at 32 768 envs; at 262 144 envs the waves of a workgroup that store in step, 8 KB at
        // a time, are 6 us of 41 faster than waves that store on their own)
        const int q0 = 192 * (tid >> 6) + (tid & 63);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int q = q0 + 64 * j;
            if (q < nq) dst[q] = stg[q];
        }
        return;
    }
    for (int q = tid; q < nq; q += T) dst[q] = stg[q];
}

// ---- the same inside the obs kernel's fused step (mg_render_kernel.h: mg_step_render_encode) ----------------------------
// A wave has its batch of kb envs staged in LDS anyway — the stepped grids (env j at raw + j * stride, HBM layout) and the
// stepped records (env j's at recs + j * rec_stride) — and the batch is cells [cell0, cell0 + kb * cells) of the flat output
// stream: the agents' marks go into the grid bytes themselves (n_obj + 4 n <= 256 is the caller's precondition) — and out again
// when the caller still needs the grids —, the stream leaves as aligned 16-byte chunks between
// a head and a tail of fewer than 16 single bytes each.  `tab`: one dword per byte value — object kinds, then the agent
// codes n_obj + 4 k + dir — (type, colour, state).  Plain functions of `lane` (of `nl` >= 32 lanes) like the phases above.
// (undo: take the marks out again — the caller still needs the grids as they were)
MG_HD void encode_batch_mark(const MgConfig& cfg, uint8_t* raw, const uint64_t* recs, int rec_stride, int kb, uint32_t m_n, int lane, int nl,
                             bool undo = false) {
    const int n = cfg.n_agents, items = kb * n;
    for (int it = lane; it < items; it += nl) {
        const uint32_t e = enc_div((uint32_t)it, (uint32_t)n, m_n);
        const int k = it - (int)e * n;
        const uint64_t* rr = recs + (size_t)e * rec_stride;
        const uint64_t r = rr[k];
        if (!(rec_byte(r, MG_AG_FLAGS) & MG_AF_PLACED)) continue;
        const uint32_t xy = rec_xy(r), rank = rec_byte(r, MG_AG_RANK);
        bool first = true;
        for (int j = 0; j < n; j++) {
            const uint64_t rj = rr[j];
            if ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy && rec_byte(rj, MG_AG_RANK) < rank) first = false;
        }
        if (!first) continue;
        const int addr = (int)e * cfg.cells_stride + (int)rec_byte(r, MG_AG_X) * cfg.H + (int)rec_byte(r, MG_AG_Y);
        const uint8_t code = (uint8_t)((uint32_t)cfg.n_obj + 4u * (uint32_t)k + rec_byte(r, MG_AG_DIR));
        if (!undo) { if (raw[addr] == 0) raw[addr] = code; }
        else if (raw[addr] == code) raw[addr] = 0;           // (object ids are < n_obj: only a mark has this value)
    }
}
MG_HD uint32_t encode_batch_byte(const uint8_t* raw, const uint32_t* tab, uint32_t s, int cells, int stride, uint32_t m_cells) {
    const uint32_t f = enc_mulhi(s, 0xAAAAAAABu) >> 1, p = s - 3u * f;          // byte s of the batch: byte p of its cell f
    const uint32_t e = enc_div(f, (uint32_t)cells, m_cells);
    return (tab[raw[(int)e * stride + (int)(f - e * (uint32_t)cells)]] >> (8u * p)) & 0xFFu;
}
struct alignas(16) EncChunk { uint32_t v[4]; };
// the aligned chunk that starts at byte s of the batch's stream
MG_HD EncChunk encode_batch_chunk(const uint8_t* raw, const uint32_t* tab, uint32_t s, int cells, int stride, int len, uint32_t m_cells) {
    const uint32_t f = enc_mulhi(s, 0xAAAAAAABu) >> 1, p = s - 3u * f;      // the chunk's first byte is byte p of cell f
    const uint32_t e = enc_div(f, (uint32_t)cells, m_cells);
    const int c = (int)f - (int)e * cells;
    const int nA = cells - c;                                               // cells left in this env, this one included
    const int addrA = (int)e * stride + c;
    // six consecutive cells: a window of the env's bytes, continued — where the env ends inside it — by the next env's first
    const bool needB = nA < 6 && (int)f + nA < len;
    const int addrB = needB ? ((int)e + 1) * stride : 0;
    MG_ENC_BOUNDS(addrA & ~3, 12);
    MG_ENC_BOUNDS(addrB, 8);
    const uint32_t sh = (uint32_t)addrA & 3u, s8 = 8u * (uint32_t)(nA < 7 ? nA : 7);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(raw + (addrA & ~3));
    const uint32_t* wb = reinterpret_cast<const uint32_t*>(raw + addrB);
    const uint64_t A = (uint64_t)enc_align(w[1], w[0], sh) | ((uint64_t)enc_align(w[2], w[1], sh) << 32);
    const uint64_t Bv = (uint64_t)wb[0] | ((uint64_t)wb[1] << 32);
    const uint64_t X = needB ? ((A & ~(~0ull << s8)) | (Bv << s8)) : A;
    uint32_t t[6];
#pragma unroll
    for (int i = 0; i < 6; i++) t[i] = tab[(uint32_t)(X >> (8 * i)) & 0xFFu];
    // 18 bytes = 6 x 3, as five dwords; the chunk is bytes p .. p + 15 of them (as in encode_chunks)
    const uint32_t d0 = enc_perm(t[1], t[0], 0x04020100u);
    const uint32_t d1 = enc_perm(t[2], t[1], 0x05040201u);
    const uint32_t d2 = enc_perm(t[3], t[2], 0x06050402u);
    const uint32_t d3 = enc_perm(t[5], t[4], 0x04020100u);
    const uint32_t d4 = enc_perm(0u, t[5], 0x0C0C0201u);
    const EncChunk o = {{enc_align(d1, d0, p), enc_align(d2, d1, p), enc_align(d3, d2, p), enc_align(d4, d3, p)}};
    return o;
}
MG_HD void encode_batch_chunks(const MgConfig& cfg, const uint8_t* raw, const uint32_t* tab, uint8_t* out, long long cell0, int kb,
                               int cells, uint32_t m_cells, int lane, int nl) {
    const int stride = cfg.cells_stride, len = kb * cells;
    const uint32_t total = 3u * (uint32_t)len;
    uint8_t* dst = out + 3 * cell0;
    uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    if (head > total) head = total;
    const uint32_t nq = (total - head) >> 4, tail0 = head + 16u * nq;
    if ((uint32_t)lane < head) dst[lane] = (uint8_t)encode_batch_byte(raw, tab, (uint32_t)lane, cells, stride, m_cells);
    if (lane >= 16 && tail0 + (uint32_t)(lane - 16) < total && lane < 32)
        dst[tail0 + (uint32_t)(lane - 16)] = (uint8_t)encode_batch_byte(raw, tab, tail0 + (uint32_t)(lane - 16), cells, stride, m_cells);
    // two chunks per lane and trip: two independent chains of look-ups in flight (a trip is three dependent LDS round trips)
    for (uint32_t q = (uint32_t)lane; q < nq; q += 2u * (uint32_t)nl) {
        const uint32_t q2 = q + (uint32_t)nl;
        const bool two = q2 < nq;
        const EncChunk a = encode_batch_chunk(raw, tab, head + 16u * q, cells, stride, len, m_cells);
        const EncChunk b = encode_batch_chunk(raw, tab, head + 16u * (two ? q2 : q), cells, stride, len, m_cells);
        *reinterpret_cast<EncChunk*>(dst + head + 16u * q) = a;
        if (two) *reinterpret_cast<EncChunk*>(dst + head + 16u * q2) = b;
    }
}

// bytes of the raw plane a piece of PC cells can span in HBM layout (+ slack for the window reads past its end)
inline int encode_raw_bytes(int cells, int stride, int PC) {
    const long long envs = cells >= PC ? 2 : (PC + cells - 2) / cells + 1;      // envs a piece can touch
    long long span = cells >= PC ? (long long)PC + (stride - cells) + 8 : envs * stride;
    span += stride < 64 ? stride : 64;                                          // the window of the env after the last
    return (int)((span + 16 + 15) / 16 * 16);
}

// everything the launch derives from the config on the host
inline EncodeLaunch encode_launch(const MgConfig& cfg, const void* out, int& PC) {
    EncodeLaunch lc;
    lc.cells = cfg.W * cfg.H;
    lc.total = (long long)cfg.B * lc.cells;
    lc.m_cells = (uint32_t)((0x100000000ull + (uint32_t)lc.cells - 1) / (uint32_t)lc.cells);
    lc.m_n = (uint32_t)((0x100000000ull + (uint32_t)cfg.n_agents - 1) / (uint32_t)cfg.n_agents);
    lc.two = cfg.n_obj + 4 * cfg.n_agents > 256 ? 1 : 0;
    lc.aligned = (reinterpret_cast<uintptr_t>(out) & 15) == 0 ? 1 : 0;
    lc.runs = (!lc.two && lc.aligned && lc.cells >= 16) ? 1 : 0;
    // pieces of 8192 cells (512 threads: the tables and the three barriers of a workgroup per 24 KB of output) where that makes 512
    // workgroups, of 4096 where that does, else of 1024 cells (one wave each)
    if (PC == 0) PC = lc.total / 8192 >= 512 ? 8192 : lc.total / 4096 >= 512 ? 4096 : 1024;
    if (PC < 1024) PC = 1024;
    lc.nraw = encode_raw_bytes(lc.cells, cfg.cells_stride, PC);
    if (lc.runs && lc.total / PC < 2048) lc.runs = 2;       // (fewer than 2048 pieces: the waves of a workgroup store on their own)
    return lc;
}

}  // namespace mg

// mg_core.h — the lane-per-env bodies of the engine (seed / reset / step / live placement) as plain
// inline functions: one call advances ONE env.  On the GPU every lane of a kernel runs one of them
// for its own env (mg_step.hip, mg_reset.hip, mg_rng.hip); the same text compiles with g++ for the
// host harness under tests/native, which steps the oracle's scenarios through these bodies on the
// CPU (a development check of the state machine — the product never loads it).
//
// Nothing here touches a GPU builtin: scratch that a kernel keeps in LDS is handed in as pointers
// with an element stride (`S`: the workgroup size on the device — column `tid` of an [n][S] array —
// and 1 on the host).
#pragma once

#include <math.h>
#include <stdint.h>

#include "marlgrid_hip.h"

#if defined(__HIPCC__)
#define MG_HD __host__ __device__ __forceinline__
#else
#define MG_HD inline
#endif
// A 32-bit value made opaque to the compiler AT THIS POINT (device code; nothing on the host): what is computed from it
// stays behind this point — the compiler otherwise hoists a float64 division out of the rarely taken branch that needs
// it, and starts on loaded values (with the wait that takes) the moment they are requested.
#if defined(__HIP_DEVICE_COMPILE__)
#define MG_OPAQUE32(x) asm volatile("" : "+v"(x))
#else
#define MG_OPAQUE32(x) do {} while (0)
#endif

namespace mg {

// ---- packed agent record (include/marlgrid_hip.h MG_AG_*) -------------------------------------
MG_HD uint32_t rec_byte(uint64_t r, int i) { return (uint32_t)(r >> (8 * i)) & 0xFFu; }
MG_HD uint64_t rec_set(uint64_t r, int i, uint32_t v) {
    return (r & ~(0xFFull << (8 * i))) | ((uint64_t)(v & 0xFFu) << (8 * i));
}
MG_HD uint32_t rec_xy(uint64_t r) { return (uint32_t)r & 0xFFFFu; }  // x | y<<8

// A per-env runtime error (the reference's exception): the first one sticks in error[b]; the one-word summary
// MgState.error_flag — usually host-mapped memory a host polls without synchronising — is raised with a
// system-scope atomic (error path only: nothing on the normal path touches it).
MG_HD void record_error(const MgState& st, int b, int err) {
    if (!err) return;
    if (st.error[b] == 0) st.error[b] = err;
    if (st.error_flag) {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_or(st.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
        *st.error_flag |= 1;
#endif
    }
}

// forward vector per dir: agents.py:183  [(1,0),(0,1),(-1,0),(0,-1)]
MG_HD int dir_dx(int d) { return d == 0 ? 1 : (d == 2 ? -1 : 0); }
MG_HD int dir_dy(int d) { return d == 1 ? 1 : (d == 3 ? -1 : 0); }

// ---- per-env MT19937: lazy regeneration + a look-ahead head ------------------------------------
// numpy's RandomState regenerates all 624 words when a block is exhausted.  The same sequence
// falls out of regenerating word `pos` only when it is needed (x[k+624] = f(x[k], x[k+1], x[k+397])
// reads exactly the values the in-place block loop reads), which needs no 624-iteration twist stall
// in one lane.  State per env: w[624] with `pos` = the next word to regenerate, and `head` = the
// MG_MT_HEAD tempered outputs x[G-16 .. G) that were generated last and are not consumed yet (G the
// generation count, pos = G mod 624).  A kernel draws from the head — a contiguous 64 B per env that
// is read with the rest of the env's record, so the shuffle of a step needs no dependent memory
// round trip; a kernel that needs more than 16 draws (placements, resets) regenerates the head 16
// words at a time, and when it is done it tops the head up again (mt_finish) — every refill with all
// of its loads in flight together.
MG_HD uint32_t mt_temper(uint32_t v) {
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}
MG_HD uint32_t mt_twist(uint32_t wi, uint32_t wi1, uint32_t wim) {
    const uint32_t y = (wi & 0x80000000u) | (wi1 & 0x7fffffffu);
    return wim ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Regenerate the next `cnt` (<= 8) words in place and return them tempered.  Every operand is read
// before anything is written: word i+1 must be read in its OLD state (the loop form reads it before
// regenerating it), and x[k+397] of a batch of 8 never lies inside the batch.
MG_HD void mt_generate(uint32_t* w, int& pos, int cnt, uint32_t* out) {
    uint32_t a[8], b[8], c[8];
    int at[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < cnt) {
            int p = pos + i;
            if (p >= MG_MT_N) p -= MG_MT_N;
            const int p1 = (p + 1 == MG_MT_N) ? 0 : p + 1;
            int pm = p + 397;
            if (pm >= MG_MT_N) pm -= MG_MT_N;
            at[i] = p;
            a[i] = w[p]; b[i] = w[p1]; c[i] = w[pm];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < cnt) {
            const uint32_t x = mt_twist(a[i], b[i], c[i]);
            w[at[i]] = x;
            out[i] = mt_temper(x);
        }
    }
    int p = pos + cnt;
    if (p >= MG_MT_N) p -= MG_MT_N;
    pos = p;
}

// A whole head of 16 words in ONE memory round trip, through LDS-DMA (gfx950: global_load_lds_dwordx4 — the data goes from
// HBM into LDS without passing through registers): mt_generate's batch of 8 keeps its 24 operands in registers, which is
// why a head refill is two DEPENDENT round trips; a batch of 16 in registers (33 operands) spilled seventy registers of the
// obs kernel's 16-wave instantiation.  Here the operands — words p .. p+16 and p+397 .. p+412, two contiguous runs — land in
// the wave's scratch `dma` (9 rows of 128 bytes: row r holds 16 bytes for each of the 8 stepping lanes, lane l at byte 16 l:
// the LDS address of an LDS-DMA load is wave-uniform, the lane's place in the row is its lane id) and are read back one at
// a time.  Why it matters: a reset draws ~70 words on ONE lane, i.e. four or five refills, and every other lane of the
// wave — and the launch, which lasts as long as its slowest wave — waits for each of them (profiles/r04/README.md
// section 3: a step in which a few envs finish 0.207 -> 0.195 ms, the step in which all do 0.36 -> 0.31 ms).  Returns
// false, having done nothing, when a run would wrap around the end of the state (5 % of the refills: the caller takes the
// register path).  (Requesting the NEXT batch's operands right after a refill, so that a chain of refills waits for
// memory once, was measured too: correct, ten spilled registers instead of two, and slower — profiles/r04.)
constexpr int kMtDmaBufDwords = 9 * 32;        // the landing zone: 9 rows of 128 bytes
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool mt_generate16_dma(uint32_t* w, int& pos, uint32_t* head, int hstride, uint32_t* dma, int dcol) {
    const int p = pos;
    const int pm = p + 397 >= MG_MT_N ? p + 397 - MG_MT_N : p + 397;
    if (p + 20 > MG_MT_N || pm + 16 > MG_MT_N) return false;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
    for (int i = 0; i < 5; i++) __builtin_amdgcn_global_load_lds((gptr)(w + p + 4 * i), (lptr)(dma + 32 * i), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_amdgcn_global_load_lds((gptr)(w + pm + 4 * i), (lptr)(dma + 32 * (5 + i)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t* mine = dma + 4 * dcol;
    uint32_t a = mine[0];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t b = mine[32 * ((i + 1) >> 2) + ((i + 1) & 3)];
        const uint32_t c = mine[32 * (5 + (i >> 2)) + (i & 3)];
        const uint32_t x = mt_twist(a, b, c);
        w[p + i] = x;
        head[i * hstride] = mt_temper(x);
        a = b;
    }
    pos = p + 16 >= MG_MT_N ? p + 16 - MG_MT_N : p + 16;
    return true;
}
#endif

struct Mt {
    uint32_t* w;            // this env's 624 words (HBM)
    int pos;                // next word to regenerate
    uint32_t* head;         // head word i at head[i * hstride] (HBM: stride 1; LDS column: stride S)
    int hstride;
    int used;               // draws taken since the kernel started
    uint32_t* dma = nullptr;   // the obs kernel's fused step: the wave's LDS landing zone for mt_generate16_dma (null: none) ...
    int dcol = 0;              // ... and this lane's column in it
    bool ahead_regs = true;    // mt_ahead may keep its operands in registers (false — the obs kernel, which runs at its register
                               // limit —: through the landing zone or not at all)

    // The head is consumed as a ring: draw number `used` is head[used % 16]; when a whole head has
    // been consumed and more is needed (placements, resets) the next 16 outputs are generated into it
    // in two batches of 8 — two memory round trips per 16 draws instead of one per draw.
    MG_HD uint32_t next() {
        const int r = used & (MG_MT_HEAD - 1);
        if (r == 0 && used != 0) {
            bool refilled = false;
#if defined(__HIP_DEVICE_COMPILE__)
            if (dma) refilled = mt_generate16_dma(w, pos, head, hstride, dma, dcol);
#endif
            if (!refilled) {
#pragma unroll
                for (int h = 0; h < MG_MT_HEAD; h += 8) {
                    uint32_t t[8];
                    mt_generate(w, pos, 8, t);
#pragma unroll
                    for (int i = 0; i < 8; i++) head[(h + i) * hstride] = t[i];
                }
            }
        }
        used++;
        return head[r * hstride];
    }
    // numpy legacy masked rejection (RandomState.randint with array bounds / shuffle's
    // random_interval): smallest 2^k-1 >= max; redraw until (w & mask) <= max; max==0 draws nothing
    MG_HD uint32_t bounded(uint32_t max) {
        if (max == 0) return 0;
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
};

// The operands of the head's top-up, requested AHEAD: a step that only shuffles has drawn all it will draw long
// before it ends (2-3 words for three agents), and the top-up at its end is three dependent HBM loads per word —
// 2 us that every wave of the launch spent waiting in mt_finish.  Right after the shuffle the step asks for the
// operands of the `used` (<= kMtAhead) words it has consumed — w[pos + i], w[pos + i + 1], w[pos + i + 397] —; they
// travel while the agents act, and mt_finish takes them if nothing has been drawn or regenerated since (no
// placement, no reset: the state's pos and the draw count are what they were).
constexpr int kMtAhead = 4;     // (mt_ahead's loads are written for exactly 4)
struct MtAhead {
    uint32_t a[kMtAhead + 1], c[kMtAhead];
    int pos, used;          // the state these operands belong to (used == 0: nothing requested)
    bool in_lds;            // the obs kernel: the operands were sent to the wave's LDS landing zone (LDS-DMA: rows 0 - 2 of Mt::dma,
                            // 16 bytes each per stepping lane) instead of nine registers that would stay live — and, at the
                            // kernel's 128-register limit, be spilled with a full wait right after the request — across the
                            // agents' resolution; mt_ahead_fetch collects them when they are used
};
MG_HD MtAhead mt_ahead(const Mt& mt) {
    // (three loads — 16 + 4 + 16 bytes — and nothing else: the step is a chain of dependent instructions in which
    // every instruction counts; as nine conditional loads with their own wrap-around arithmetic the request cost
    // what it saved.  A `pos` whose operands wrap around the end of the state is left to mt_finish's own loads.)
    MtAhead ah;
    const int p = mt.pos;
    int pm = p + 397;
    if (pm >= MG_MT_N) pm -= MG_MT_N;
    ah.pos = p;
    ah.used = (mt.used >= 1 && mt.used <= kMtAhead && p + kMtAhead < MG_MT_N && pm + kMtAhead <= MG_MT_N) ? mt.used : 0;
    ah.in_lds = false;
    if (ah.used) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (mt.dma) {
            typedef const __attribute__((address_space(1))) void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
            // (three 16-byte rows like mt_generate16_dma's: w[p .. p+3], w[p+4 .. p+7] — only its first word is an operand —,
            // w[pm .. pm+3]; a 4-byte LDS-DMA load for the one word landed somewhere else on gfx950: measured, one env in 30 000)
            if (p + 8 > MG_MT_N) { ah.used = 0; return ah; }
            __builtin_amdgcn_global_load_lds((gptr)(mt.w + p), (lptr)(mt.dma), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(mt.w + p + 4), (lptr)(mt.dma + 32), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(mt.w + pm), (lptr)(mt.dma + 64), 16, 0, 0);
            ah.in_lds = true;
            return ah;
        }
#endif
        if (!mt.ahead_regs) { ah.used = 0; return ah; }
        __builtin_memcpy(&ah.a[0], mt.w + p, 16);
        ah.a[4] = mt.w[p + 4];
        __builtin_memcpy(&ah.c[0], mt.w + pm, 16);
    }
    return ah;
}
// the operands as mt_finish uses them: the registers mt_ahead filled, or — after ONE wait for the wave's outstanding
// loads — what it sent to the LDS landing zone
MG_HD MtAhead mt_ahead_fetch(const Mt& mt, const MtAhead& ah) {
    MtAhead r = ah;
#if defined(__HIP_DEVICE_COMPILE__)
    if (ah.in_lds) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t* z = mt.dma + 4 * mt.dcol;
#pragma unroll
        for (int i = 0; i < kMtAhead; i++) {
            // (opaque: as plain loads the compiler merges "this LDS word or that field of *ah" into a load through a phi of
            // POINTERS — and a struct whose fields' addresses flow into a phi lives in scratch memory, all of StepCtx with it)
            uint32_t va = z[i], vc = z[64 + i];
            MG_OPAQUE32(va); MG_OPAQUE32(vc);
            r.a[i] = va; r.c[i] = vc;
        }
        uint32_t v4 = z[32];
        MG_OPAQUE32(v4);
        r.a[kMtAhead] = v4;
    }
#endif
    (void)mt;
    return r;
}

// After the env's last draw of the kernel: what is left of the current head slides down to the front,
// the rest is generated, so that the head again holds the next 16 outputs.  `head_out`: the env's 16
// head words in HBM (may be the array `mt.head` points at: the copy runs upwards, dst < src).
MG_HD void mt_finish(Mt& mt, uint32_t* head_out, const MtAhead* ah = nullptr) {
    if (mt.used == 0) return;
    const int r = mt.used & (MG_MT_HEAD - 1);
    const int k = r ? r : MG_MT_HEAD;           // words consumed from the current head
    const int keep = MG_MT_HEAD - k;
    for (int j = 0; j < keep; j++) head_out[j] = mt.head[(j + k) * mt.hstride];
    if (ah && ah->used && ah->used == mt.used && ah->pos == mt.pos) {      // (k == used <= kMtAhead)
        int p = mt.pos;
        const MtAhead got = mt_ahead_fetch(mt, *ah);
#pragma unroll
        for (int i = 0; i < kMtAhead; i++) {
            if (i < k) {
                uint32_t a0 = got.a[i], a1 = got.a[i + 1], c0 = got.c[i];
                MG_OPAQUE32(a0); MG_OPAQUE32(a1); MG_OPAQUE32(c0);      // (used HERE, not where they were requested)
                const uint32_t x = mt_twist(a0, a1, c0);
                mt.w[p] = x;
                head_out[keep + i] = mt_temper(x);
                if (++p == MG_MT_N) p = 0;
            }
        }
        mt.pos = p;
        mt.used = 0;
        return;
    }
    for (int j = keep; j < MG_MT_HEAD; j += 8) {
        const int cnt = (MG_MT_HEAD - j) < 8 ? (MG_MT_HEAD - j) : 8;
        uint32_t t[8];
        mt_generate(mt.w, mt.pos, cnt, t);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < cnt) head_out[j + i] = t[i];
    }
    mt.used = 0;
}

// The same for a caller that writes the head back ITSELF (the obs kernel: the whole wave copies its batch's heads to
// HBM as contiguous runs, instead of sixteen scattered stores per stepping lane): the new outputs go into the slots of
// the consumed ones, mt.head becomes a ring, and the returned k is where it starts — the head in order is
// mt.head[((j + k) & 15) * hstride], j = 0 .. 15.  The state words and `pos` are written as in mt_finish.
MG_HD int mt_finish_ring(Mt& mt, const MtAhead* ah = nullptr) {
    if (mt.used == 0) return 0;
    const int r = mt.used & (MG_MT_HEAD - 1);
    const int k = r ? r : MG_MT_HEAD;           // words consumed from the current head: slots 0 .. k-1
    if (ah && ah->used && ah->used == mt.used && ah->pos == mt.pos) {      // (k == used <= kMtAhead)
        int p = mt.pos;
        const MtAhead got = mt_ahead_fetch(mt, *ah);
#pragma unroll
        for (int i = 0; i < kMtAhead; i++) {
            if (i < k) {
                uint32_t a0 = got.a[i], a1 = got.a[i + 1], c0 = got.c[i];
                MG_OPAQUE32(a0); MG_OPAQUE32(a1); MG_OPAQUE32(c0);
                const uint32_t x = mt_twist(a0, a1, c0);
                mt.w[p] = x;
                mt.head[i * mt.hstride] = mt_temper(x);
                if (++p == MG_MT_N) p = 0;
            }
        }
        mt.pos = p;
    } else {
        for (int j = 0; j < k; j += 8) {
            const int cnt = (k - j) < 8 ? (k - j) : 8;
            uint32_t t[8];
            mt_generate(mt.w, mt.pos, cnt, t);
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < cnt) mt.head[(j + i) * mt.hstride] = t[i];
        }
    }
    mt.used = 0;
    return k & (MG_MT_HEAD - 1);
}

// MultiGridEnv.seed -> gym seeding.np_random -> RandomState.seed([k0(, k1)]) = init_by_array
// (base.py:371-374), then the first head.
MG_HD void mt_seed_env(const uint32_t* key, int klen, uint32_t* mt, int32_t* mt_pos, uint32_t* head) {
    if (klen < 1) klen = 1;
    if (klen > MG_KEY_WORDS) klen = MG_KEY_WORDS;
    uint32_t prev = 19650218u;   // init_genrand(19650218)
    mt[0] = prev;
    for (int i = 1; i < MG_MT_N; i++) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mt[i] = prev;
    }
    int i = 1, j = 0;
    prev = mt[0];
    for (int k = MG_MT_N; k; k--) {   // max(N, key_len) == N for key_len <= 2
        const uint32_t kj = (j == 0) ? key[0] : key[1];
        const uint32_t v = (mt[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + kj + (uint32_t)j;
        mt[i] = v;
        prev = v;
        i++; j++;
        if (i >= MG_MT_N) { mt[0] = prev; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = MG_MT_N - 1; k; k--) {
        const uint32_t v = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[i] = v;
        prev = v;
        i++;
        if (i >= MG_MT_N) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
    // numpy's pos == 624 (nothing generated yet) is pos 0 of the lazy form; then the first head
    int pos = 0;
    for (int h = 0; h < MG_MT_HEAD; h += 8) mt_generate(mt, pos, 8, head + h);
    *mt_pos = pos;
}

// ---- placement helpers ---------------------------------------------------------------------------
// agents of this env standing on cell xy (x | y << 8); records in an [n][S] array, column `col`
MG_HD int agents_on(const uint64_t* rec, int S, int col, int n, uint32_t xy) {
    int cnt = 0;
    for (int j = 0; j < n; j++) {
        const uint64_t rj = rec[j * S + col];
        cnt += ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy) ? 1 : 0;
    }
    return cnt;
}
// arrival: agent k gets the highest rank, everyone above its old rank slides down (the reference's
// `obj.agents.append`, base.py:547-552 / 684-686)
MG_HD uint64_t arrive(uint64_t* rec, int S, int col, int n, uint64_t r) {
    const uint32_t old_rank = rec_byte(r, MG_AG_RANK);
    for (int j = 0; j < n; j++) {
        const uint64_t rj = rec[j * S + col];
        const uint32_t rk = rec_byte(rj, MG_AG_RANK);
        if (rk > old_rank) rec[j * S + col] = rec_set(rj, MG_AG_RANK, rk - 1);
    }
    return rec_set(r, MG_AG_RANK, (uint32_t)(n - 1));
}

// place_obj(agent, **agent_spawn_kwargs) (base.py:690-708 over try_place_obj :664-688) for agent k,
// which is off the grid: rejection-sample a cell of the spawn rectangle whose object can be
// overlapped (or that is empty) and, without ghost_mode, holds no agent.  `keep_flags`: flag bits
// that survive (a live re-seat keeps DONE).  Returns false when max_tries draws all failed.
MG_HD bool place_agent(const MgConfig& cfg, const uint8_t* oflags, uint8_t* g, Mt& mt, uint64_t* rec, int S, int col,
                       int k, uint32_t keep_flags) {
    const int n = cfg.n_agents, H = cfg.H;
    uint64_t r = rec[k * S + col];
    const int x0 = cfg.spawn_x0, y0 = cfg.spawn_y0;
    const uint32_t mx = (uint32_t)(cfg.spawn_x1 - x0 - 1), my = (uint32_t)(cfg.spawn_y1 - y0 - 1);
    for (int t = 0; t < cfg.spawn_max_tries; t++) {
        // np_random.randint(top, bottom): low + bounded(high - low - 1) per coordinate
        const int x = x0 + (int)mt.bounded(mx);
        const int y = y0 + (int)mt.bounded(my);
        if (cfg.spawn_reject && cfg.spawn_reject[x * H + y]) continue;   // reject_fn(pos): a spent try (base.py:700-701)
        const uint32_t base = g[x * H + y];
        const uint32_t xy = (uint32_t)x | ((uint32_t)y << 8);
        const int cnt = agents_on(rec, S, col, n, xy);
        if ((base == 0 || (oflags[base] & MG_OF_CAN_OVERLAP)) && (cnt == 0 || (cfg.ghost_mode & 2))) {
            r = arrive(rec, S, col, n, r);
            r = rec_set(r, MG_AG_X, (uint32_t)x);
            r = rec_set(r, MG_AG_Y, (uint32_t)y);
            r = rec_set(r, MG_AG_FLAGS, (rec_byte(r, MG_AG_FLAGS) & keep_flags) | MG_AF_ACTIVE | MG_AF_PLACED);
            rec[k * S + col] = r;
            return true;
        }
    }
    return false;
}

// ---- reset (base.py:402-416) ----------------------------------------------------------------------
// `_gen_grid` as a static template + ordered rejection-sampled placements (base.py:664-708;
// envs/cluttered.py:25-36, envs/empty.py:9-16, envs/goalcycle.py:30-51), then agent.reset(
// new_episode=True) (agents.py:161-170; dir survives) and place_obj + activate in index order
// (base.py:409-412).  Works on the env's records in rec[.][col] (read: old dir; written: the new
// records) and on its grid slice `g` in HBM; the caller stores the records and the counters.
MG_HD int reset_env(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, const uint8_t* oflags, int b,
                    uint8_t* g, Mt& mt, uint64_t* rec, int S, int col) {
    const int n = cfg.n_agents, H = cfg.H;
    {   // self.grid = MultiGrid(...); wall_rect; put_obj — the static part of _gen_grid (16-byte vectors)
        typedef struct { uint32_t v[4]; } __attribute__((aligned(16))) q16;
        const q16* src = reinterpret_cast<const q16*>(prog.template_grid);
        q16* dst = reinterpret_cast<q16*>(g);
        for (int i = 0; i < cfg.cells_stride / 16; i++) dst[i] = src[i];
    }
    int err = 0;
    // place_obj(obj, max_tries) for non-agent objects: only an empty cell accepts (try_place_obj,
    // base.py:669-679; no agent is on the fresh grid yet)
    for (int o = 0; o < prog.n_ops && !err; o++) {
        const MgGenOp op = prog.ops[o];
        if (op.max_tries == 0) {      // a static edit after a placement (put_obj / wall helper): replaces what is there
            for (int x = op.x0; x < op.x1; x++)
                for (int y = op.y0; y < op.y1; y++) g[x * H + y] = (uint8_t)op.obj;
            continue;
        }
        for (int c = 0; c < op.count && !err; c++) {
            bool ok = false;
            for (int t = 0; t < op.max_tries; t++) {
                const int x = op.x0 + (int)mt.bounded((uint32_t)(op.x1 - op.x0 - 1));
                const int y = op.y0 + (int)mt.bounded((uint32_t)(op.y1 - op.y0 - 1));
                const int cell = x * H + y;
                if (op.reject >= 0 && prog.reject[(size_t)op.reject * cfg.cells_stride + cell]) continue;   // reject_fn(pos)
                if (g[cell] == 0) { g[cell] = (uint8_t)op.obj; ok = true; break; }
            }
            if (!ok) err = MG_ERR_RECURSION;
        }
    }
    for (int k = 0; k < n; k++) {
        uint64_t nr = 0;
        nr = rec_set(nr, MG_AG_DIR, rec_byte(rec[k * S + col], MG_AG_DIR) & 3u);
        nr = rec_set(nr, MG_AG_RANK, (uint32_t)k);
        nr = rec_set(nr, MG_AG_BONUS, 0xFFu);
        rec[k * S + col] = nr;
        if (cfg.prestige_mask) st.prestige[(size_t)b * n + k] = 0.0;   // new_episode=True: agents.py:167-168
    }
    // ranks 0..n-1 in index order, and `arrive` keeps them a permutation: an agent placed now gets
    // the highest rank, i.e. stacks above every agent placed before it, as the reference's append does
    for (int k = 0; k < n && !err; k++) {
        if (cfg.spawn_delay[k] != 0) continue;    // later spawns happen in step (base.py:503-506)
        if (!place_agent(cfg, oflags, g, mt, rec, S, col, k, 0)) err = MG_ERR_RECURSION;
    }
    return err;
}

// ---- step (base.py:501-649) -------------------------------------------------------------------------
// Per env the reference is strictly sequential: agents act in a freshly shuffled order
// (base.py:514-516) and each action sees the grid left by the previous one.
//
// Flat state model that reproduces the reference's object graph: a cell holds a
// base object id (0 = none) and any number of agents; the ordered stack the reference keeps in
// `obj.agents` lists (append on entry base.py:547-552, remove on exit :555-559, re-seat left-behind
// agents in order :562-569) is exactly "agents in this cell sorted by arrival", carried as a rank
// permutation: a successful move gives the mover the highest rank.
struct StepScratch {        // per-workgroup arrays, this env is column `col`, element stride S
    uint64_t* rec;          // [n][S] agent records
    uint32_t* head;         // [MG_MT_HEAD][S] look-ahead RNG words
    uint8_t* act;           // [n][S] action (0xFF: invalid)
    uint8_t* fb;            // [n][S] front-cell base id, pre-loaded (null: `g` is cheap to read, no pre-load)
    const MgObjDesc* obj;   // [n_obj] object table (shared)
    const uint8_t* oflags;  // [n_obj] object flags (shared)
    int S, col;
    bool defer_writeback = false;   // the caller writes records and RNG head back itself (StepOut::head_k; the obs kernel)
    uint32_t* dma = nullptr;        // the obs kernel: the wave's LDS landing zone for one-round-trip head refills (mt_generate16_dma)
    // the obs kernel's agent-parallel resolution (step_par_*, S = 8): [n][8] flags / turns, the envs' step counts, and
    // where step_par_commit puts the settled records (a second [n][8] column set: the sequential loop of an env that
    // needs it still reads the old ones in `rec`)
    uint8_t* ord = nullptr;         // [n][S] iter_order of an env with more than 16 agents (step_begin)
    uint8_t* pflag = nullptr;
    uint8_t* ordp = nullptr;
    int32_t* psc = nullptr;
    uint64_t* rec_out = nullptr;
#if defined(MG_AB_VARIANTS)
    unsigned long long* stamp = nullptr;   // measurement build: 5 words, wall_clock64 at the section ends of step_run (or null)
#endif
};
struct StepEnv { int pos0, sc0; };
// what step_run reports: whether it wrote the grid slice; with StepScratch::defer_writeback also where the RNG head
// ring starts (mt_finish_ring) — the stepped records are in StepScratch::rec, the head in StepScratch::head
struct StepOut { bool wrote; int head_k; };
#if defined(MG_AB_VARIANTS) && defined(__HIP_DEVICE_COMPILE__)
#define MG_STEP_STAMP(i) do { if (sc.stamp) sc.stamp[i] = wall_clock64(); } while (0)
#else
#define MG_STEP_STAMP(i) do {} while (0)
#endif

// round trip 1: everything whose address is known up front, all of it contiguous across the batch.
// actions: [B][n] little-endian integers of `action_bytes` (1, 4 or 8) bytes each.
MG_HD StepEnv step_load(const MgConfig& cfg, const MgState& st, const void* actions, int action_bytes, int b,
                        const StepScratch& sc) {
    const int n = cfg.n_agents, S = sc.S, col = sc.col;
    for (int k = 0; k < n; k++) sc.rec[k * S + col] = st.agents[(size_t)b * n + k];
    for (int k = 0; k < n; k++) {
        const size_t i = (size_t)b * n + k;
        const long long a = action_bytes == 8 ? (long long)static_cast<const int64_t*>(actions)[i]
                          : action_bytes == 4 ? (long long)static_cast<const int32_t*>(actions)[i]
                                              : (long long)static_cast<const uint8_t*>(actions)[i];
        sc.act[k * S + col] = (a >= 0 && a <= 6) ? (uint8_t)a : (uint8_t)0xFF;
    }
    for (int i = 0; i < MG_MT_HEAD; i++) sc.head[i * S + col] = st.mt_head[(size_t)b * MG_MT_HEAD + i];
    StepEnv e;
    e.pos0 = st.mt_pos[b];
    e.sc0 = st.step_count[b];
    return e;
}

// What entering the cell object `fbase` does to the agent that just moved onto it (base.py:576-585): the reward of a Goal
// / BonusTile (hasattr(fwd_cell, 'get_reward'); float64, decayed like base.py:579), the agent's bonus state, done on a
// Goal / Lava.  `r`: the mover's record with its new position; returns it with bonus state and flags updated.
struct MoveEffect { float rew; bool rewarded; double rwd_applied; };
MG_HD uint64_t enter_cell(const MgConfig& cfg, const MgObjDesc* obj, uint32_t fbase, uint32_t fflags, uint64_t r, uint32_t flags,
                          int step_count, MoveEffect& fx) {
    const MgObjDesc od = obj[fbase];
    if (od.reward_kind) {                 // hasattr(fwd_cell,'get_reward') :576-581
        double rwd;
        if (od.reward_kind == 1) {
            rwd = od.reward;              // Goal.get_reward objects.py:219-220
        } else {                          // BonusTile.get_reward objects.py:180-206
            int bs = (int)rec_byte(r, MG_AG_BONUS);
            bool first_bonus = false;
            const int nb = od.n_bonus ? od.n_bonus : 1;
            if (bs == 0xFF) { bs = ((int)od.bonus_id - 1 + nb) % nb; first_bonus = true; }
            if (bs == od.bonus_id) rwd = -fabs(od.penalty);
            else if ((bs + 1) % nb == od.bonus_id) { bs = od.bonus_id; rwd = od.reward; }
            else rwd = -fabs(od.penalty);
            if (od.bonus_flags & 2) bs = od.bonus_id;
            if (first_bonus && !(od.bonus_flags & 1)) rwd = 0.0;
            r = rec_set(r, MG_AG_BONUS, (uint32_t)bs);
        }
        // reward decay factor, float64 like the reference (base.py:579) — worked out where a reward is paid (a float64
        // division: a hundred dependent cycles that most steps of most envs do not need)
        int t = step_count;
        MG_OPAQUE32(t);
        const double decay = cfg.reward_decay ? (1.0 - 0.9 * ((double)t / (double)cfg.max_steps)) : 1.0;
        fx.rwd_applied = rwd * decay;
        fx.rewarded = true;
        fx.rew = (float)fx.rwd_applied;
    }
    if (fflags & MG_OF_ENDS_EPISODE) r = rec_set(r, MG_AG_FLAGS, flags | MG_AF_DONE);  // :584-585
    return r;
}

// auto_reset: an env whose episode ends in this step starts its next one right away (`prog`; reset
// fused into the step: the done flag still reports the end).
// `g`: the env's grid slice — its home in HBM (st.grid + b * cells_stride), or a staged copy of it (the obs
// kernel steps the envs it is about to render on their LDS copies: no dependent HBM round trip per grid
// look-up); returns whether the slice was written (the owner of a staged copy then writes it back).
//
// A step in three parts, so that the obs kernel can put its agent-parallel resolution (step_par_*, below) between the
// first and the last: step_begin (late spawns, step_count, the shuffle), step_agents (the sequential action loop),
// step_end (done agents, respawn, episode end, auto-reset, write-back).  step_run = the three in a row.
struct StepCtx {            // what an env's step carries from part to part (one lane's registers)
    Mt mt;
    uint64_t order;         // iter_order: agent of turn i in nibble i
    int err, step_count;
    MtAhead ahead;
    bool grid_dirty;
};

MG_HD StepCtx step_begin(const MgConfig& cfg, const MgState& st, int b, const StepEnv& env, const StepScratch& sc, uint8_t* g) {
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H, S = sc.S, col = sc.col;
    uint64_t* s_rec = sc.rec;
    StepCtx c;
    c.mt = Mt{st.mt + (size_t)b * MG_MT_N, env.pos0, sc.head + col, S, 0};
    c.mt.dma = sc.dma;
    c.mt.dcol = col;
    c.mt.ahead_regs = !sc.defer_writeback;       // (defer_writeback: the obs kernel's fused step)
    c.err = 0;
    c.grid_dirty = false;
    MG_STEP_STAMP(0);

    // late spawns (base.py:503-506), before step_count is incremented and before the shuffle: any agent
    // that is neither active nor done (spawn_delay not reached at reset, or lifted off the grid by a
    // failed live placement) is placed as soon as step_count >= its spawn_delay
    for (int k = 0; k < n; k++) {
        const uint32_t f = rec_byte(s_rec[k * S + col], MG_AG_FLAGS);
        if (!(f & (MG_AF_ACTIVE | MG_AF_DONE)) && env.sc0 >= cfg.spawn_delay[k])
            if (!place_agent(cfg, sc.oflags, g, c.mt, s_rec, S, col, k, 0)) c.err = c.err ? c.err : MG_ERR_RECURSION;
    }

    // round trip 2: every agent's front cell.  An agent's position and heading are only ever changed
    // by its own action, so its front cell is known before the loop; the cell's *content* can only be
    // changed by a pickup / drop / toggle earlier in this step (grid_dirty), in which case it is re-read.
    // (sc.fb == nullptr: `g` is a staged copy in LDS — the obs kernel's fused step —, every look-up is as cheap as
    // one into `fb` and the pre-load would only be three more LDS round trips per agent)
    if (sc.fb)
        for (int k = 0; k < n; k++) {
            const uint64_t r = s_rec[k * S + col];
            const int dir = (int)rec_byte(r, MG_AG_DIR);
            const int fx = (int)rec_byte(r, MG_AG_X) + dir_dx(dir), fy = (int)rec_byte(r, MG_AG_Y) + dir_dy(dir);
            const bool ok = (rec_byte(r, MG_AG_FLAGS) & MG_AF_ACTIVE) && fx >= 0 && fx < W && fy >= 0 && fy < H;
            sc.fb[k * S + col] = ok ? g[fx * H + fy] : (uint8_t)0;
        }

    c.step_count = env.sc0 + 1;   // base.py:512

    // iter_order = arange(n); np_random.shuffle(iter_order)  (base.py:514-516): legacy Fisher-Yates
    // over numpy's masked-rejection bounded draws — served from the look-ahead head
    // (up to 16 agents the permutation is sixteen nibbles of one register: a swap is four shifts instead of four LDS round
    // trips; with more agents — up to MG_MAX_AGENTS — it lives in the step's scratch column sc.ord)
    static_assert(MG_MAX_AGENTS <= 256, "iter_order: one byte per agent in sc.ord");
    // (the identity, made HERE: as a loop invariant of the obs kernel's env loop it is hoisted, spilled at the kernel's
    // register limit and fetched back from scratch memory in front of every shuffle)
    uint32_t id_lo = 0x76543210u, id_hi = 0xFEDCBA98u;
    MG_OPAQUE32(id_lo); MG_OPAQUE32(id_hi);
    uint64_t order = (uint64_t)id_lo | ((uint64_t)id_hi << 32);
    if (n <= 16) {
        for (int i = n - 1; i >= 1; i--) {
            const int j = (int)c.mt.bounded((uint32_t)i);
            const uint64_t oi_ = (order >> (4 * i)) & 0xFull, oj_ = (order >> (4 * j)) & 0xFull;
            order = (order & ~((0xFull << (4 * i)) | (0xFull << (4 * j)))) | (oj_ << (4 * i)) | (oi_ << (4 * j));
        }
    } else {
        for (int i = 0; i < n; i++) sc.ord[i * S + col] = (uint8_t)i;
        for (int i = n - 1; i >= 1; i--) {
            const int j = (int)c.mt.bounded((uint32_t)i);
            const uint8_t t = sc.ord[i * S + col];
            sc.ord[i * S + col] = sc.ord[j * S + col];
            sc.ord[j * S + col] = t;
        }
    }
    c.order = order;
    c.ahead = mt_ahead(c.mt);
    MG_STEP_STAMP(1);
    return c;
}

MG_HD void step_agents(const MgConfig& cfg, const MgState& st, float* rewards, int b, const StepScratch& sc, uint8_t* g, StepCtx& c) {
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H, S = sc.S, col = sc.col;
    uint64_t* s_rec = sc.rec;
    const bool direct = !sc.fb;     // (no pre-loaded front cells: the action loop reads `g`)
    int err = c.err;
    bool grid_dirty = c.grid_dirty;
    for (int oi = 0; oi < n; oi++) {
        const int k = n <= 16 ? (int)((c.order >> (4 * oi)) & 0xFull) : (int)sc.ord[oi * S + col];
        MoveEffect fxm = {0.0f, false, 0.0};      // agent.reward(rwd) was called: `rewarded` (prestige bookkeeping)
        uint64_t r = s_rec[k * S + col];
        const uint32_t flags = rec_byte(r, MG_AG_FLAGS);
        if (flags & MG_AF_ACTIVE) {   // base.py:521
            const int action = (int)sc.act[k * S + col];
            const int cx = (int)rec_byte(r, MG_AG_X), cy = (int)rec_byte(r, MG_AG_Y);
            const int dir = (int)rec_byte(r, MG_AG_DIR);
            const int fx = cx + dir_dx(dir), fy = cy + dir_dy(dir);   // agent.front_pos agents.py:194-198
            if (fx < 0 || fx >= W || fy < 0 || fy >= H) {
                err = err ? err : MG_ERR_ASSERT;   // grid.get asserts (base.py:154-156)
            } else {
                const int fcell = fx * H + fy;
                const uint32_t fbase = (direct || grid_dirty) ? (uint32_t)g[fcell] : (uint32_t)sc.fb[k * S + col];
                const uint32_t fxy = (uint32_t)fx | ((uint32_t)fy << 8);
                const uint32_t fflags = sc.oflags[fbase];
                if (action == 0) {                                   // left  base.py:530-531
                    r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 3) & 3));
                } else if (action == 1) {                            // right :534-535
                    r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 1) & 3));
                } else if (action == 2 || action == 4) {             // forward :538-585 / drop :600-606
                    const int agents_there = agents_on(s_rec, S, col, n, fxy);
                    if (action == 2) {
                        // fwd_cell is None, or it can_overlap(); the top object is the base object if
                        // there is one, else the first agent standing there (agents overlap)
                        bool can_move = fbase ? (fflags & MG_OF_CAN_OVERLAP) != 0 : true;
                        if (!(cfg.ghost_mode & 1) && fbase == 0 && agents_there > 0) can_move = false;  // :541-542
                        if (can_move && (flags & MG_AF_EVICTED)) {
                            // the agent is in no cell (put_obj replaced the one it stood on): "remove agent from old
                            // cell" (base.py:555-559) fails — the assert on a solid object, list.remove -> ValueError
                            // on an overlappable object or on another agent, AttributeError on None
                            const uint32_t cb = g[cx * H + cy];
                            const int e2 = cb ? ((sc.oflags[cb] & MG_OF_CAN_OVERLAP) ? MG_ERR_VALUE : MG_ERR_ASSERT)
                                              : (agents_on(s_rec, S, col, n, (uint32_t)cx | ((uint32_t)cy << 8)) ? MG_ERR_VALUE : MG_ERR_ATTRIBUTE);
                            err = err ? err : e2;
                        } else if (can_move) {
                            r = arrive(s_rec, S, col, n, r);
                            r = rec_set(r, MG_AG_X, (uint32_t)fx);
                            r = rec_set(r, MG_AG_Y, (uint32_t)fy);
                            if (fbase) r = enter_cell(cfg, sc.obj, fbase, fflags, r, flags, c.step_count, fxm);
                        }
                    } else {
                        // drop: `if not fwd_cell and agent.carrying`
                        const uint32_t carry = rec_byte(r, MG_AG_CARRY);
                        if (fbase == 0 && agents_there == 0 && carry) {
                            g[fcell] = (uint8_t)carry;
                            grid_dirty = true;
                            r = rec_set(r, MG_AG_CARRY, 0);
                        }
                    }
                } else if (action == 3) {                            // pickup :590-597
                    if (fbase && (fflags & MG_OF_CAN_PICKUP) && rec_byte(r, MG_AG_CARRY) == 0) {
                        r = rec_set(r, MG_AG_CARRY, fbase);
                        g[fcell] = 0;
                        grid_dirty = true;
                    }
                } else if (action == 5) {                            // toggle :609-613
                    if (fbase) {
                        if (fflags & MG_OF_IS_BOX) {
                            err = err ? err : MG_ERR_TYPE;           // Box.toggle arity objects.py:381-382
                        } else if (fflags & MG_OF_IS_DOOR) {         // Door.toggle objects.py:333-346
                            const MgObjDesc od = sc.obj[fbase];
                            if (fflags & MG_OF_DOOR_LOCKED) {
                                const uint32_t carry = rec_byte(r, MG_AG_CARRY);
                                if (carry) {
                                    const MgObjDesc cd = sc.obj[carry];
                                    if ((cd.flags & MG_OF_IS_KEY) && cd.color_idx == od.color_idx)
                                        { g[fcell] = od.unlock_next; grid_dirty = true; }
                                }
                            } else {
                                g[fcell] = od.toggle_next;
                                grid_dirty = true;
                            }
                        }
                    }
                } else if (action == 6) {                            // done :616-617
                } else {
                    err = err ? err : MG_ERR_VALUE;                  // :619-620
                }
            }
            s_rec[k * S + col] = r;
            if (cfg.prestige_mask) {
                // agent.reward(rwd) then agent.on_step(): agents.py:141-153 (allow_negative_prestige=False)
                double* pp = st.prestige + (size_t)b * n + k;
                double p = *pp;
                if (fxm.rewarded) p = (fxm.rwd_applied >= 0) ? p + fxm.rwd_applied : 0.0;
                *pp = p * cfg.prestige_beta[k];
            }
        }
        rewards[(size_t)b * n + k] = fxm.rew;
    }
    c.err = err;
    c.grid_dirty = grid_dirty;
    MG_STEP_STAMP(2);
}

MG_HD StepOut step_end(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, bool auto_reset, int b,
                       const StepScratch& sc, uint8_t* g, StepCtx& c) {
    const int n = cfg.n_agents, S = sc.S, col = sc.col;
    uint64_t* s_rec = sc.rec;
    Mt& mt = c.mt;
    int err = c.err, step_count = c.step_count;
    bool grid_dirty = c.grid_dirty;
    // done agents (base.py:627-646), in index order: without respawn they are deactivated but stay
    // where they are; with respawn they leave their cell (an agent only ever becomes done on a Goal /
    // Lava, i.e. inside that object's stack, so nothing is left behind), drop what they carry
    // (agent.reset(new_episode=False), agents.py:161-166) and are re-placed by rejection sampling
    // among the agents currently on the grid.  Then episode done (base.py:649).
    bool all_done = true;
    for (int k = 0; k < n; k++) {
        uint64_t r = s_rec[k * S + col];
        const uint32_t f = rec_byte(r, MG_AG_FLAGS);
        if (f & MG_AF_DONE) {
            if (cfg.respawn) {
                r = rec_set(r, MG_AG_FLAGS, 0);
                r = rec_set(r, MG_AG_CARRY, 0);
                s_rec[k * S + col] = r;                      // off the grid while sampling
                if (!place_agent(cfg, sc.oflags, g, mt, s_rec, S, col, k, 0)) err = err ? err : MG_ERR_RECURSION;
                all_done = false;
            } else {
                s_rec[k * S + col] = rec_set(r, MG_AG_FLAGS, f & ~MG_AF_ACTIVE);
            }
        } else all_done = false;
    }
    const bool done = (step_count >= cfg.max_steps) || all_done;
    if (done && auto_reset) {
        const int e2 = reset_env(cfg, st, prog, sc.oflags, b, g, mt, s_rec, S, col);
        err = err ? err : e2;
        step_count = 0;
        grid_dirty = true;
    }
    MG_STEP_STAMP(3);
    int head_k = -1;
    if (sc.defer_writeback) {
        head_k = mt_finish_ring(mt, &c.ahead);
    } else {
        for (int k = 0; k < n; k++) st.agents[(size_t)b * n + k] = s_rec[k * S + col];
        mt_finish(mt, st.mt_head + (size_t)b * MG_MT_HEAD, &c.ahead);
    }
    st.step_count[b] = step_count;
    st.mt_pos[b] = mt.pos;
    st.done[b] = (uint8_t)done;
    record_error(st, b, err);
    MG_STEP_STAMP(4);
    StepOut out = {grid_dirty, head_k};
    return out;
}

MG_HD StepOut step_run(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, bool auto_reset, float* rewards,
                       int b, const StepEnv& env, const StepScratch& sc, uint8_t* g) {
    StepCtx c = step_begin(cfg, st, b, env, sc, g);
    step_agents(cfg, st, rewards, b, sc, g, c);
    return step_end(cfg, st, prog, auto_reset, b, sc, g, c);
}

// ---- the action loop resolved by ONE LANE PER AGENT (the obs kernel's fused step) ---------------------------------------
// The reference's loop is sequential per env (base.py:517-622), and so is step_agents: ~1 700 dependent instructions at
// three agents — LDS round trips, mostly — on ONE lane, at the head of every wave of the obs kernel.  But what an agent does
// rarely depends on what the agents before it did: a turn never does; a forward move with ghost_mode (the default) depends
// on the cell OBJECT in front — which only a pickup / drop / toggle changes — and not on who stands there; the only thing
// the order always decides is the arrival RANK (who lies on top of whom), and that has a closed form: the agents that did
// not move keep their relative order at the bottom, the movers follow in the order they moved (= `arrive` applied in turn).
// So, between step_begin and step_end, the wave resolves the agents of its (<= 8) staged envs with one lane per (agent,
// env): lane l = 8 k + j is agent k of staged env j (S = 8 columns, n <= 8 agents).
//   step_par_resolve  reads the PRE-step state and either settles the agent's action or asks for the sequential loop for
//                     its env — conservatively: anything that writes the grid (a pickup, drop or toggle that would take
//                     effect), any error (front cell outside the grid, an unknown action, an evicted agent that moves),
//                     and, without ghost_mode, a forward move onto an empty cell that another agent occupies or also
//                     moves onto;
//   step_par_commit   works out the ranks and, unless some agent of the env asked for the loop, writes the records, the
//                     rewards and 'prestige' — else it writes nothing and the env's lane runs step_agents as before.
// Bit-exact by construction: an env either takes a path on which no agent's outcome depends on the order (ranks aside,
// and those are computed from the order), or it takes the sequential loop.  Both halves are plain functions of `lane` that
// talk through the step's LDS columns only (no cross-lane builtins), so that tests/native runs them lane by lane on the host.
struct ParLane {            // what a lane carries from step_par_resolve to step_par_commit
    uint64_t r;             // the agent's record after its action (rank not yet)
    MoveEffect fx;
    bool live, active, moved;
};
// sc.pflag [n][8] u8: bit 0 the agent moved, bit 1 its env needs the sequential loop; sc.ordp [n][8] u8: the agent's turn;
// sc.psc [8] i32: the env's step_count (after the increment)
MG_HD void step_par_publish(const MgConfig& cfg, const StepScratch& sc, const StepCtx& c) {      // the env's lane, after step_begin
    const int n = cfg.n_agents, col = sc.col;
    for (int i = 0; i < n; i++) sc.ordp[(int)((c.order >> (4 * i)) & 0xFull) * 8 + col] = (uint8_t)i;
    sc.psc[col] = c.step_count;
}
MG_HD ParLane step_par_resolve(const MgConfig& cfg, const StepScratch& sc, const uint8_t* grids, int kb, int lane) {
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H;
    const int j = lane & 7, k = lane >> 3;
    ParLane P;
    P.fx = MoveEffect{0.0f, false, 0.0};
    P.live = k < n && j < kb;
    P.active = P.moved = false;
    P.r = 0;
    if (!P.live) return P;
    const uint8_t* g = grids + (size_t)j * cfg.cells_stride;
    uint64_t r = sc.rec[k * 8 + j];
    const uint32_t flags = rec_byte(r, MG_AG_FLAGS);
    bool serial = false;
    if (flags & MG_AF_ACTIVE) {
        P.active = true;
        const int action = (int)sc.act[k * 8 + j];
        const int cx = (int)rec_byte(r, MG_AG_X), cy = (int)rec_byte(r, MG_AG_Y), dir = (int)rec_byte(r, MG_AG_DIR);
        const int fx = cx + dir_dx(dir), fy = cy + dir_dy(dir);
        if (fx < 0 || fx >= W || fy < 0 || fy >= H || action > 6) {
            serial = true;                                            // AssertionError / ValueError: the loop records them
        } else {
            const uint32_t fbase = g[fx * H + fy], fflags = sc.oflags[fbase];
            const uint32_t fxy = (uint32_t)fx | ((uint32_t)fy << 8);
            if (action == 0) r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 3) & 3));
            else if (action == 1) r = rec_set(r, MG_AG_DIR, (uint32_t)((dir + 1) & 3));
            else if (action == 2) {
                const bool can_move = fbase ? (fflags & MG_OF_CAN_OVERLAP) != 0 : true;
                if (can_move && (flags & MG_AF_EVICTED)) serial = true;
                if (can_move && !(cfg.ghost_mode & 1) && fbase == 0) {
                    // blocked iff an agent stands there WHEN THIS AGENT ACTS (base.py:541-542): nobody there now and nobody
                    // heading there -> free whatever the order; anything else is the loop's business
                    for (int i = 0; i < n; i++) {
                        if (i == k) continue;
                        const uint64_t ri = sc.rec[i * 8 + j];
                        const uint32_t fi = rec_byte(ri, MG_AG_FLAGS);
                        if ((fi & MG_AF_PLACED) && rec_xy(ri) == fxy) serial = true;
                        if ((fi & MG_AF_ACTIVE) && sc.act[i * 8 + j] == 2) {
                            const int di = (int)rec_byte(ri, MG_AG_DIR);
                            const uint32_t txy = (uint32_t)((int)rec_byte(ri, MG_AG_X) + dir_dx(di)) | ((uint32_t)((int)rec_byte(ri, MG_AG_Y) + dir_dy(di)) << 8);
                            if (txy == fxy) serial = true;
                        }
                    }
                }
                if (can_move && !serial) {
                    P.moved = true;
                    r = rec_set(r, MG_AG_X, (uint32_t)fx);
                    r = rec_set(r, MG_AG_Y, (uint32_t)fy);
                    if (fbase) r = enter_cell(cfg, sc.obj, fbase, fflags, r, flags, sc.psc[j], P.fx);
                }
            } else if (action == 3) {
                if (fbase && (fflags & MG_OF_CAN_PICKUP) && rec_byte(r, MG_AG_CARRY) == 0) serial = true;
            } else if (action == 4) {
                if (fbase == 0 && rec_byte(r, MG_AG_CARRY)) serial = true;       // (whoever stands there: the loop decides)
            } else if (action == 5) {
                if (fbase && (fflags & (MG_OF_IS_BOX | MG_OF_IS_DOOR))) serial = true;
            }
        }
    }
    P.r = r;
    sc.pflag[k * 8 + j] = (uint8_t)((P.moved ? 1 : 0) | (serial ? 2 : 0));
    return P;
}
// returns whether the lane's env needs the sequential loop (the same answer in every lane of the env)
MG_HD bool step_par_commit(const MgConfig& cfg, const MgState& st, float* rewards, int b0, const StepScratch& sc, const ParLane& P, int lane) {
    const int n = cfg.n_agents;
    const int j = lane & 7, k = lane >> 3;
    if (!P.live) return false;
    // ranks (`arrive`, base.py:547-552 / 684-686, applied in turn order): non-movers keep their order at the bottom, the
    // movers lie on top of them in the order of their turns
    const uint32_t my_rank = rec_byte(sc.rec[k * 8 + j], MG_AG_RANK), my_turn = sc.ordp[k * 8 + j];
    uint32_t serial = 0, below = 0, stay = 0, before = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t f = sc.pflag[i * 8 + j];
        serial |= f & 2u;
        const bool mv = (f & 1u) != 0;
        const uint32_t ri = (uint32_t)reinterpret_cast<const uint8_t*>(sc.rec + i * 8 + j)[MG_AG_RANK];
        stay += mv ? 0u : 1u;
        below += (!mv && ri < my_rank) ? 1u : 0u;
        before += (mv && sc.ordp[i * 8 + j] < my_turn) ? 1u : 0u;
    }
    if (serial) return true;
    const size_t at = (size_t)(b0 + j) * n + k;
    uint64_t r = rec_set(P.r, MG_AG_RANK, P.moved ? stay + before : below);
    if (P.active && cfg.prestige_mask) {
        // agent.reward(rwd) then agent.on_step(): agents.py:141-153 (allow_negative_prestige=False)
        double p = st.prestige[at];
        if (P.fx.rewarded) p = (P.fx.rwd_applied >= 0) ? p + P.fx.rwd_applied : 0.0;
        st.prestige[at] = p * cfg.prestige_beta[k];
    }
    rewards[at] = P.fx.rew;
    sc.rec_out[k * 8 + j] = r;
    return false;
}

// ---- MultiGridEnv.reset for one env (explicit reset; the auto-reset runs inside step_run) ---------
MG_HD void reset_run(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, const uint8_t* oflags, int b,
                     bool clear_done, uint64_t* rec, int S, int col) {
    const int n = cfg.n_agents;
    uint8_t* g = st.grid + (size_t)b * cfg.cells_stride;
    uint32_t* head = st.mt_head + (size_t)b * MG_MT_HEAD;
    Mt mt{st.mt + (size_t)b * MG_MT_N, st.mt_pos[b], head, 1, 0};
    for (int k = 0; k < n; k++) rec[k * S + col] = st.agents[(size_t)b * n + k];
    const int err = reset_env(cfg, st, prog, oflags, b, g, mt, rec, S, col);
    for (int k = 0; k < n; k++) st.agents[(size_t)b * n + k] = rec[k * S + col];
    mt_finish(mt, head);
    st.mt_pos[b] = mt.pos;
    st.step_count[b] = 0;
    if (clear_done) st.done[b] = 0;
    record_error(st, b, err);
}

// ---- MultiGridEnv.place_obj / try_place_obj on a live grid (base.py:664-708) for one env -------------
MG_HD void place_run(const MgConfig& cfg, const MgState& st, const uint8_t* oflags, int b, int what, int x0, int y0,
                     int x1, int y1, int max_tries, const int32_t* fixed_pos, const uint8_t* reject, int32_t* out_pos,
                     uint8_t* out_ok, uint64_t* rec, int S, int col) {
    const int n = cfg.n_agents, H = cfg.H;
    uint8_t* g = st.grid + (size_t)b * cfg.cells_stride;
    uint32_t* head = st.mt_head + (size_t)b * MG_MT_HEAD;
    Mt mt{st.mt + (size_t)b * MG_MT_N, st.mt_pos[b], head, 1, 0};
    for (int k = 0; k < n; k++) rec[k * S + col] = st.agents[(size_t)b * n + k];
    const bool is_agent = what < 0;
    const int k = -(what + 1);
    if (is_agent) {   // off the grid while a cell is looked for
        const uint64_t r = rec[k * S + col];
        rec[k * S + col] = rec_set(r, MG_AG_FLAGS, rec_byte(r, MG_AG_FLAGS) & ~(MG_AF_PLACED | MG_AF_ACTIVE));
    }
    bool ok = false;
    int x = -1, y = -1;
    const int tries = fixed_pos ? 1 : max_tries;
    for (int t = 0; t < tries && !ok; t++) {
        if (fixed_pos) { x = fixed_pos[2 * b]; y = fixed_pos[2 * b + 1]; if (x < 0 || x >= cfg.W || y < 0 || y >= H) break; }
        else {
            x = x0 + (int)mt.bounded((uint32_t)(x1 - x0 - 1));
            y = y0 + (int)mt.bounded((uint32_t)(y1 - y0 - 1));
            if (reject && reject[x * H + y]) continue;       // reject_fn(pos): a spent try (base.py:700-701)
        }
        const uint32_t base = g[x * H + y];
        const uint32_t xy = (uint32_t)x | ((uint32_t)y << 8);
        const int cnt = agents_on(rec, S, col, n, xy);
        if (!is_agent) {
            // only an empty cell (no object, no agent) accepts a non-agent object (base.py:672-679)
            if (base == 0 && cnt == 0) { g[x * H + y] = (uint8_t)what; ok = true; }
        } else if ((base == 0 || (oflags[base] & MG_OF_CAN_OVERLAP)) && (cnt == 0 || (cfg.ghost_mode & 2))) {
            uint64_t r = arrive(rec, S, col, n, rec[k * S + col]);
            r = rec_set(r, MG_AG_X, (uint32_t)x);
            r = rec_set(r, MG_AG_Y, (uint32_t)y);
            r = rec_set(r, MG_AG_FLAGS, (rec_byte(r, MG_AG_FLAGS) & MG_AF_DONE) | MG_AF_ACTIVE | MG_AF_PLACED);
            rec[k * S + col] = r;
            ok = true;
        }
    }
    for (int j = 0; j < n; j++) st.agents[(size_t)b * n + j] = rec[j * S + col];
    mt_finish(mt, head);
    st.mt_pos[b] = mt.pos;
    if (out_pos) { out_pos[2 * b] = ok ? x : -1; out_pos[2 * b + 1] = ok ? y : -1; }
    if (out_ok) out_ok[b] = ok ? 1 : 0;
    if (!ok && !fixed_pos) record_error(st, b, MG_ERR_RECURSION);
}

}  // namespace mg

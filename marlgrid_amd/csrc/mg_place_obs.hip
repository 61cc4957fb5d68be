// mg_place_obs.hip — mg_obs_place / mg_obs_release / mg_obs_trim (include/marlgrid_hip.h): WHERE in HBM the observation
// buffers of a batch live.  Construction-time host code: nothing on the step path allocates or calls in here.
//
// What is measured, not modelled (profiles/r04/README.md section 1, profiles/r05/README.md): the obs raster's write
// pattern — thousands of concurrent sequential streams — runs 20-25 % faster into a buffer that straddles the boundary
// between two of the driver's physical blocks than into one that lies inside a block (one valley per allocation, at the
// block boundary, +-half a buffer wide; a dense fill runs at the same speed anywhere).  The driver builds an allocation
// from power-of-two blocks, largest first, so a candidate is ONE hipMalloc of 3 P bytes (P = the power of two >= half the
// buffer: a 2 P block followed by a P block) and the buffer is the window centred on the 2 P | P junction.  Whether the
// junction is a boundary that pays is the one thing that has to be timed — the raster itself, `iters` launches between
// two HIP events.  Candidates are drawn, and kept alive so that the driver moves on through its free lists, until
// `n_buffers` of them run `gain` under the median candidate; then every other candidate goes back to the driver.
//
// The one piece of process state in this library lives here: the record of the arenas handed out (mg_obs_release has to
// find the allocation a window belongs to) and of released arenas of the fast class, which the next mg_obs_place of the
// same size on the same device takes instead of searching again.  Guarded by a mutex.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

#include "mg_device.h"
#include "mg_launch.h"

namespace {

struct Arena {
    void* base;            // what hipMalloc returned
    uint64_t bytes;        // ... and how much
    uint64_t offset;       // the buffer: [base + offset, base + offset + buffer_bytes)
    uint64_t buffer_bytes;
    int device;
    float ms;              // the raster into it when it was chosen
    bool fast;             // it was chosen as one of the fast class (found = 1)
    int32_t geom[5];       // the raster it was timed with: B, viewers, view_size, tile_size, n_agents — `ms` means nothing to another
};

std::mutex g_mu;
std::vector<Arena> g_out;       // handed to callers (mg_obs_place -> mg_obs_release)
std::vector<Arena> g_spare;     // released, fast class: the next placement of the same size takes these first

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// average duration (ms) of `iters` raster launches into `obs` (one more, untimed, in front)
hipError_t time_raster(const MgConfig& cfg, const MgState& st, uint8_t* obs, int iters, hipStream_t s, hipEvent_t t0, hipEvent_t t1,
                       float* ms) {
    hipError_t err = mg::launch_render(cfg, st, obs, nullptr, nullptr, nullptr, s);
    (void)hipEventRecord(t0, s);
    for (int i = 0; i < iters && err == hipSuccess; i++) err = mg::launch_render(cfg, st, obs, nullptr, nullptr, nullptr, s);
    (void)hipEventRecord(t1, s);
    (void)hipEventSynchronize(t1);
    float v = 0.f;
    (void)hipEventElapsedTime(&v, t0, t1);
    *ms = v / (float)iters;
    return err;
}

struct Cand {
    void* base;
    uint64_t bytes, offset;
    float ms;
};

constexpr double kFastRate = 5.9e12;     // bytes per second the raster writes into a buffer of the fast class (and above)

uint64_t spare_cap(int device) {
    size_t fr = 0, total = 0;
    (void)device;
    if (hipMemGetInfo(&fr, &total) != hipSuccess) return 0;
    return std::min<uint64_t>(8ull << 30, (uint64_t)total / 16);
}

}  // namespace

extern "C" {

int32_t mg_obs_place(const MgConfig* cfg, const MgState* st, int32_t n_buffers, uint64_t budget_bytes, double seconds,
                     int32_t flags, const MgPlaceTuning* tuning, void** out, MgPlaceStats* stats, void* stream) {
    if (!cfg || !st || !out || n_buffers < 1 || n_buffers > MG_PLACE_MAX || !cfg->obj || !cfg->atlas || cfg->B < 1) return MG_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nv = cfg->n_view ? cfg->n_view : cfg->n_agents, P = cfg->view_size * cfg->tile_size;
    const uint64_t nbytes = (uint64_t)cfg->B * nv * P * P * 3;
    MgPlaceTuning tn = {};
    if (tuning) tn = *tuning;
    const double gain = tn.gain > 0 ? tn.gain : 0.12;
    // (a candidate costs 1-2 ms where memory has been used before: the time limit is the bound that matters; a box on which the
    // default search drew its 64 candidates in 80 ms without a hit, and found its pair in the next 10, is why the default is 192)
    const int max_cands = tn.max_candidates > 0 ? tn.max_candidates : 192;
    const int iters = tn.iters > 0 ? tn.iters : 3;
    const uint64_t min_bytes = tn.min_bytes ? tn.min_bytes : (256ull << 20);
    const double slow_alloc = tn.slow_alloc_s_per_gib > 0 ? tn.slow_alloc_s_per_gib : (tn.slow_alloc_s_per_gib < 0 ? 0.0 : 0.02);
    const uint64_t stir_cap = tn.stir_bytes ? tn.stir_bytes : (64ull << 30);
    const double fast_rate = tn.fast_rate > 0 ? tn.fast_rate : (tn.fast_rate < 0 ? 1e30 : kFastRate);
    const bool default_seconds = seconds <= 0;
    if (default_seconds) seconds = 2.0;
    const uint64_t share = tn.share > 1 ? (uint64_t)tn.share : 1;     // this process counts on 1 / share of what is free
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return MG_E_LAUNCH;
    MgPlaceStats S = {};
    S.buffer_bytes = nbytes;
    const double t_begin = now_s();
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MG_E_LAUNCH;

    const int keep = n_buffers;
    const int32_t geom[5] = {cfg->B, nv, cfg->view_size, cfg->tile_size, cfg->n_agents};
    std::vector<Cand> kept;                 // what goes to the caller
    bool fast = false;
    hipError_t err = hipSuccess;

    // 1. released arenas of the fast class, this size, this device: measured once, taken
    if (!(flags & MG_PLACE_NO_REUSE)) {
        std::vector<Arena> mine;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (size_t i = 0; i < g_spare.size() && (int)mine.size() < keep;)
                // (same size AND same raster: an arena's `ms` was measured under the configuration that released it — another view /
                // tile / agent split of the same byte count writes another pattern at another speed, and neither "still as fast
                // as it was" nor "fast class" would mean anything for it)
                if (g_spare[i].device == device && g_spare[i].buffer_bytes == nbytes && std::equal(geom, geom + 5, g_spare[i].geom)) { mine.push_back(g_spare[i]); g_spare.erase(g_spare.begin() + i); }
                else i++;
        }
        for (const Arena& a : mine) {
            float ms = 0.f;
            err = time_raster(*cfg, *st, (uint8_t*)a.base + a.offset, iters, s, e0, e1, &ms);
            S.windows++;
            if (err == hipSuccess && ms <= 1.06f * a.ms) kept.push_back({a.base, a.bytes, a.offset, ms});
            else (void)hipFree(a.base);     // (its neighbourhood changed: not what it was)
        }
        S.reused = (int32_t)kept.size();
        if ((int)kept.size() == keep) fast = true;
    }

    // 2. small buffers: plain allocations (the effect needs thousands of concurrent streams)
    if ((int)kept.size() < keep && nbytes < min_bytes) {
        while ((int)kept.size() < keep) {
            void* p = nullptr;
            if (hipMalloc(&p, nbytes) != hipSuccess) { (void)hipGetLastError(); S.stopped = MG_PLACE_STOP_OOM; break; }
            kept.push_back({p, nbytes, 0, 0.f});
        }
        if ((int)kept.size() == keep) S.stopped = MG_PLACE_STOP_SMALL;
    }

    // 3. the search
    if ((int)kept.size() < keep && nbytes >= min_bytes) {
        const int need = keep - (int)kept.size();
        const uint64_t half = (nbytes + 1) / 2;
        uint64_t P0 = 1ull << 21;
        while (P0 < half) P0 <<= 1;                                   // the power of two >= half the buffer (>= 2 MiB)
        S.candidate_bytes = 3 * P0;
        // (the default time limit follows the candidate: memory nobody had before is cleared as it is handed out, ~0.02 s per GiB —
        // a 24 GiB candidate of BASELINE configs[4] is 0.5-0.8 s of hipMalloc + measurement: 2 s would end the search after
        // three, 4 s after six — one box needed more than six: a third of a second per GiB of candidate, 12 s at most)
        if (default_seconds) seconds = std::min(12.0, std::max(2.0, 0.33 * (double)(3 * P0) / (double)(1ull << 30)));
        const int max_level = (flags & MG_PLACE_THOROUGH) ? 2 : 0;
        const int passes = (flags & MG_PLACE_THOROUGH) ? 2 : 1;
        std::vector<Cand> cands;              // every candidate measured (base == nullptr: gone back to the driver)
        uint64_t alive = 0;
        bool stirred = false;
        auto measure = [&](void* base, uint64_t off) -> float {
            float ms = 1e30f;
            const hipError_t e = time_raster(*cfg, *st, (uint8_t*)base + off, iters, s, e0, e1, &ms);
            if (e != hipSuccess) err = e;
            S.windows++;
            return ms;
        };
        auto median_ms = [&]() -> float {
            std::vector<float> c;
            for (const Cand& k : cands) c.push_back(k.ms);
            std::sort(c.begin(), c.end());
            return c.empty() ? 0.f : c[c.size() / 2];
        };
        auto nth_ms = [&](int i) -> float {      // the i-th fastest candidate that is still allocated
            std::vector<float> c;
            for (const Cand& k : cands) if (k.base) c.push_back(k.ms);
            std::sort(c.begin(), c.end());
            return (int)c.size() > i ? c[i] : 1e30f;
        };
        auto drop_losers = [&]() {              // all but the best `need` go back to the driver (their ms stay on record)
            const float cut = nth_ms(need - 1);
            int have = 0;
            alive = 0;
            for (Cand& k : cands) {
                if (!k.base) continue;
                if (k.ms <= cut && have < need) { have++; alive += k.bytes; }
                else { (void)hipFree(k.base); k.base = nullptr; }
            }
        };
        for (int pass = 1; pass <= passes && !fast && err == hipSuccess; pass++) {
            S.passes = pass;
            double t_end = now_s() + seconds;
            int misses = 0, level = 0;
            bool plain = cands.empty();         // the first `need` candidates: plain buffer-sized allocations (the baseline)
            bool plain_stage = false, drew_since_drop = false;
            int drops = 0;
            int n_plain0 = 0;
            S.stopped = MG_PLACE_STOP_CAP;
            while ((int)cands.size() < need + max_cands * pass) {
                const bool baseline = plain && !plain_stage;
                // (no time limit before the baseline is in: the `need` plain allocations are what is kept when nothing else is found)
                if (!baseline && now_s() > t_end) { S.stopped = MG_PLACE_STOP_TIME; break; }
                const uint64_t Pk = P0 << level;
                const uint64_t arena = plain ? nbytes : 3 * Pk;
                size_t fr = 0, total = 0;
                if (hipMemGetInfo(&fr, &total) != hipSuccess) { err = hipErrorUnknown; break; }
                // live candidates: a quarter of what is free (and was free before this search took its share), 32 GiB at most — but
                // never less than the kept buffers plus THREE first-level candidates while that is within half of what is free: a
                // bound that cannot hold one candidate next to the baseline (16 GB buffers: 24 GiB candidates; round 5's 32 GiB) is
                // not a bounded search, it is no search.  `share` (MgPlaceTuning): ranks that share a device count on their part.
                const uint64_t fs = ((uint64_t)fr + alive) / share;
                const uint64_t budget = budget_bytes ? budget_bytes
                    : std::max<uint64_t>(std::min<uint64_t>(fs / 4, 32ull << 30), std::min<uint64_t>(fs / 2, (uint64_t)need * nbytes + 3 * (3 * P0)));
                S.budget_bytes = budget;
                const bool is_short = alive + arena > std::max<uint64_t>(budget, (uint64_t)need * nbytes) || arena + (256ull << 20) > fr;
                if (!baseline && (misses >= 12 || is_short) && !plain_stage) {
                    // A run of candidates that all miss: the free lists these two block sizes come from hold nothing across a
                    // boundary for now.  THOROUGH: one big allocate-and-free when the allocations were slow (memory nobody had
                    // before, cleared as it is handed out, front to back), then larger block pairs; always, at the end: plain
                    // allocations of the buffer's own size (another set of free lists; no overhead when one is kept).
                    misses = 0;
                    if ((flags & MG_PLACE_STIR) && !stirred && !is_short && S.alloc_bytes &&
                        S.alloc_seconds / (double)S.alloc_bytes >= slow_alloc / (double)(1ull << 30)) {
                        const double t0 = now_s();
                        const uint64_t big = std::min<uint64_t>(fr / 2, stir_cap);
                        void* p = nullptr;
                        if (hipMalloc(&p, big) == hipSuccess) { (void)hipFree(p); S.stirred_bytes = big; }
                        else (void)hipGetLastError();
                        stirred = true;
                        t_end += now_s() - t0;
                        continue;
                    }
                    if (level < max_level && !is_short) { level++; S.level = std::max(S.level, level); }
                    else { plain = plain_stage = true; S.plain_stage = 1; }
                    continue;
                }
                if (is_short && !baseline) {
                    // the budget is reached: the losers go back to the driver and the search goes on inside the same bound (what comes
                    // back is handed out again, but in another order and cut differently: measured, the next few draws often hit) —
                    // up to six times; a budget that does not hold one candidate next to the kept ones ends the search
                    if (drops >= 6 || (drops > 0 && !drew_since_drop)) { S.stopped = MG_PLACE_STOP_MEMORY; break; }
                    drop_losers();
                    drops++;
                    drew_since_drop = false;
                    plain = plain_stage = false;        // (back to the constructed candidates: they hit more often than plain ones)
                    misses = level = 0;
                    continue;
                }
                const double t0 = now_s();
                void* base = nullptr;
                if (hipMalloc(&base, arena) != hipSuccess) { (void)hipGetLastError(); S.stopped = MG_PLACE_STOP_OOM; break; }
                S.alloc_seconds += now_s() - t0;
                S.alloc_bytes += arena;
                alive += arena;
                drew_since_drop = true;
                const uint64_t centre = plain ? 0 : (2 * Pk - nbytes / 2) & ~4095ull;   // the window centred on the 2 P | P junction
                Cand pick = {base, arena, centre, measure(base, centre)};
                const float med = cands.empty() ? pick.ms : median_ms();
                if (!plain) {
                    // Memory that was never allocated before is handed out front to back: the two blocks of a candidate are then
                    // neighbours, the junction no boundary at all, and the boundary the search will walk across lies anywhere in
                    // some candidate: after four misses in a row the other window positions are measured too ...
                    const bool scanned = misses >= 4 && pick.ms > (1.0 - gain) * med;
                    if (scanned) {
                        const uint64_t step = std::max<uint64_t>(4096, nbytes / 5) & ~4095ull;
                        for (uint64_t o = 0; o + nbytes <= arena; o += step) {
                            if ((o > centre ? o - centre : centre - o) < step / 2) continue;
                            const float ms = measure(base, o);
                            if (ms < pick.ms) { pick.ms = ms; pick.offset = o; }
                        }
                    }
                    // ... and a window that is partly across a boundary (the gain is in proportion to the smaller share) is moved
                    // until it is centred
                    const bool partial = pick.ms > (1.0 - gain) * med;
                    if (pick.ms <= 0.96 * med && (partial || scanned)) {
                        uint64_t step = std::max<uint64_t>(4096, nbytes / 10) & ~4095ull;
                        for (int r = 0; r < 4; r++) {
                            const uint64_t at = pick.offset;
                            for (int sgn = -1; sgn <= 1; sgn += 2) {
                                if (sgn < 0 && at < step) continue;
                                const uint64_t o = sgn < 0 ? at - step : at + step;
                                if (o + nbytes > arena) continue;
                                const float ms = measure(base, o);
                                if (ms < pick.ms) { pick.ms = ms; pick.offset = o; }
                            }
                            step = std::max<uint64_t>(4096, step / 2) & ~4095ull;
                        }
                    }
                }
                cands.push_back(pick);
                if (baseline && ++n_plain0 >= need) {
                    plain = false;                                     // the baseline is in: on to the constructed candidates
                    // ... unless this configuration's raster is nowhere near what HBM takes even inside a block (5.3 TB/s): a
                    // launch that writes its buffer at under 3.5 TB/s is bound by something else — instruction issue, latency —
                    // and no placement changes that (measured: views 11 / 13 at 5-pixel tiles, 194 candidates without a class)
                    float best0 = 1e30f;
                    for (const Cand& k : cands) best0 = std::min(best0, k.ms);
                    if ((double)nbytes / ((double)best0 * 1e-3) < 3.5e12) { S.stopped = MG_PLACE_STOP_UNBOUND; break; }
                }
                if (err != hipSuccess) break;
                const float m2 = median_ms();
                misses = pick.ms <= (1.0 - gain) * m2 ? 0 : misses + 1;
                if ((int)cands.size() >= need + 4 && nth_ms(need - 1) <= (1.0 - gain) * m2) {
                    S.stopped = MG_PLACE_STOP_FOUND;
                    fast = true;
                    break;
                }
                // ... or the kept set already takes the raster's bytes at the rate the fast class reaches (5.9 TB/s: 0.150 ms for
                // the bench shard's 925 MB against 0.188 inside a block): buffers of several GB span the driver's blocks wherever
                // they lie — every candidate is "fast", none is 12 % under the median, and there is nothing left to look for
                // (measured: BASELINE configs[2] 2.5 GB buffers at 6.4-6.5 TB/s, configs[4] 16 GB at 6.2-6.4, plain or constructed)
                if ((int)cands.size() >= need && (double)nbytes / ((double)nth_ms(need - 1) * 1e-3) >= fast_rate) {
                    S.stopped = MG_PLACE_STOP_FOUND;
                    fast = true;
                    break;
                }
            }
            if (S.stopped == MG_PLACE_STOP_UNBOUND) break;     // (nothing to find: no second pass either)
            if (!fast && pass < passes) drop_losers();   // (a second pass starts from the best of the first: the free lists are in another order now)
        }
        // the best `need` of what is still allocated are kept; the rest goes back to the driver
        drop_losers();
        S.candidates = (int32_t)cands.size();
        S.median_ms = median_ms();
        for (size_t i = 0; i < cands.size() && i < MG_PLACE_ALL; i++) S.all_ms[i] = cands[i].ms;
        std::vector<Cand> best;
        for (const Cand& k : cands) if (k.base) best.push_back(k);
        std::sort(best.begin(), best.end(), [](const Cand& a, const Cand& b) { return a.ms < b.ms; });
        for (const Cand& k : best) kept.push_back(k);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);

    if ((int)kept.size() < keep || err != hipSuccess) {       // out of memory (or a failed launch): nothing is handed out
        for (const Cand& k : kept) (void)hipFree(k.base);
        S.seconds = now_s() - t_begin;
        if (stats) *stats = S;
        return err != hipSuccess ? MG_E_LAUNCH : MG_E_NOMEM;
    }
    S.found = fast ? 1 : 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int i = 0; i < keep; i++) {
            const Cand& k = kept[i];
            out[i] = (uint8_t*)k.base + k.offset;
            S.kept_ms[i] = k.ms;
            S.window_offset[i] = k.offset;
            S.arena_bytes[i] = k.bytes;
            S.pinned_bytes += k.bytes;
            Arena a = {k.base, k.bytes, k.offset, nbytes, device, k.ms, fast && k.ms > 0.f, {geom[0], geom[1], geom[2], geom[3], geom[4]}};
            g_out.push_back(a);
        }
    }
    S.seconds = now_s() - t_begin;
    if (stats) *stats = S;
    return MG_OK;
}

int32_t mg_obs_release(void* ptr) {
    if (!ptr) return MG_OK;
    Arena a = {};
    bool keep = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t i = 0;
        for (; i < g_out.size(); i++)
            if ((uint8_t*)g_out[i].base + g_out[i].offset == (uint8_t*)ptr) break;
        if (i == g_out.size()) return MG_E_ARG;          // not a buffer mg_obs_place handed out (or released twice)
        a = g_out[i];
        g_out.erase(g_out.begin() + i);
        if (a.fast) {
            uint64_t held = 0;
            int same = 0;
            for (const Arena& sp : g_spare) {
                if (sp.device == a.device) held += sp.bytes;
                if (sp.device == a.device && sp.buffer_bytes == a.buffer_bytes) same++;
            }
            int cur = 0;
            (void)hipGetDevice(&cur);
            keep = cur == a.device && same < MG_PLACE_MAX && held + a.bytes <= spare_cap(a.device);
            if (keep) g_spare.push_back(a);
        }
    }
    if (keep) return MG_OK;
    return hipFree(a.base) == hipSuccess ? MG_OK : MG_E_LAUNCH;
}

int32_t mg_obs_trim(int32_t device) {
    std::vector<Arena> gone;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_spare.size();)
            if (device < 0 || g_spare[i].device == device) { gone.push_back(g_spare[i]); g_spare.erase(g_spare.begin() + i); }
            else i++;
    }
    for (const Arena& a : gone) (void)hipFree(a.base);
    return (int32_t)gone.size();
}

}  // extern "C"

// mg_reset.hip — batched MultiGridEnv.reset (marlgrid/base.py:402-416): `_gen_grid` as a static
// template + ordered rejection-sampled placements (base.py:664-708; envs/cluttered.py:25-36,
// envs/empty.py:9-16, envs/goalcycle.py:30-51), then agent placement in index order.
//
// One lane per env (64 envs per wave): the work per env is a short, data-dependent rejection
// loop over that env's own RNG, so the batch is the only parallel axis.  The template copy is
// 16-byte vector traffic; placements are byte stores into the env's own grid slice.
#include "mg_device.h"
#include "mg_launch.h"

namespace mg {

__global__ __launch_bounds__(kBlock) void reset_kernel(MgConfig cfg, MgState st, MgGenProgram prog,
                                                       const uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_rec[];  // [n][kBlock]
    const int tid = threadIdx.x;
    const int b = blockIdx.x * kBlock + tid;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    const int n = cfg.n_agents, W = cfg.W, H = cfg.H;

    uint8_t* g = st.grid + (size_t)b * cfg.cells_stride;
    {   // self.grid = MultiGrid(...); wall_rect; put_obj  — the static part of _gen_grid
        const uint4* src = reinterpret_cast<const uint4*>(prog.template_grid);
        uint4* dst = reinterpret_cast<uint4*>(g);
        for (int i = 0; i < cfg.cells_stride / 16; i++) dst[i] = src[i];
    }
    Mt mt{st.mt + (size_t)b * MG_MT_N, st.mt_pos[b]};
    int err = 0;

    // place_obj(obj, max_tries) for non-agent objects: only an empty cell accepts (try_place_obj,
    // base.py:669-679; no agent is on the fresh grid yet)
    for (int o = 0; o < prog.n_ops && !err; o++) {
        const MgGenOp op = prog.ops[o];
        for (int c = 0; c < op.count && !err; c++) {
            bool ok = false;
            for (int t = 0; t < op.max_tries; t++) {
                // np_random.randint(top, bottom): low + bounded(high - low - 1) per coordinate
                int x = op.x0 + (int)mt.bounded((uint32_t)(op.x1 - op.x0 - 1));
                int y = op.y0 + (int)mt.bounded((uint32_t)(op.y1 - op.y0 - 1));
                int cell = x * H + y;
                if (g[cell] == 0) { g[cell] = (uint8_t)op.obj; ok = true; break; }
            }
            if (!ok) err = MG_ERR_RECURSION;
        }
    }

    // agents: agent.reset(new_episode=True) (agents.py:161-170; dir survives), then place_obj +
    // activate in index order (base.py:409-412)
    for (int k = 0; k < n; k++) {
        uint64_t r = st.agents[(size_t)b * n + k];
        uint32_t dir = rec_byte(r, MG_AG_DIR) & 3u;
        uint64_t nr = 0;
        nr = rec_set(nr, MG_AG_DIR, dir);
        nr = rec_set(nr, MG_AG_RANK, (uint32_t)k);
        nr = rec_set(nr, MG_AG_BONUS, 0xFFu);
        if (!err && cfg.spawn_delay[k] == 0) {   // later spawns happen in mg_step (base.py:503-506)
            bool ok = false;
            for (int t = 0; t < prog.agent_max_tries; t++) {
                int x = (int)mt.bounded((uint32_t)(W - 1));
                int y = (int)mt.bounded((uint32_t)(H - 1));
                uint32_t base = g[x * H + y];
                uint32_t xy = (uint32_t)x | ((uint32_t)y << 8);
                int cnt = 0;
                for (int j = 0; j < k; j++) {
                    uint64_t rj = s_rec[j * kBlock + tid];
                    cnt += ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy) ? 1 : 0;
                }
                // try_place_obj (base.py:664-688): the cell's object must can_overlap (agents do);
                // without ghost_mode an occupied cell rejects
                bool overlap_ok = (base == 0) || (cfg.obj[base].flags & MG_OF_CAN_OVERLAP);
                if (overlap_ok && (cnt == 0 || (cfg.ghost_mode & 2))) {
                    nr = rec_set(nr, MG_AG_X, (uint32_t)x);
                    nr = rec_set(nr, MG_AG_Y, (uint32_t)y);
                    nr = rec_set(nr, MG_AG_FLAGS, MG_AF_ACTIVE | MG_AF_PLACED);
                    ok = true;
                    break;
                }
            }
            if (!ok) err = MG_ERR_RECURSION;
        }
        s_rec[k * kBlock + tid] = nr;
        st.agents[(size_t)b * n + k] = nr;
        if (cfg.prestige_mask) st.prestige[(size_t)b * n + k] = 0.0;   // new_episode=True: agents.py:167-168
    }
    st.mt_pos[b] = mt.pos;
    st.step_count[b] = 0;
    // auto-reset passes the done flags themselves as the mask: they stay readable as step()'s return
    // value (the next mg_step overwrites them); an explicit reset clears them
    if (mask != st.done) st.done[b] = 0;
    if (err && st.error[b] == 0) st.error[b] = err;
}

__global__ __launch_bounds__(kBlock) void put_obj_kernel(MgConfig cfg, MgState st, int obj, int x, int y,
                                                         const uint8_t* __restrict__ mask) {
    int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    st.grid[(size_t)b * cfg.cells_stride + x * cfg.H + y] = (uint8_t)obj;
}

// MultiGridEnv.place_obj / try_place_obj on a live grid (base.py:664-708): lane per env.
__global__ __launch_bounds__(kBlock) void place_kernel(MgConfig cfg, MgState st, int what, int x0, int y0, int x1,
                                                       int y1, int max_tries, const int32_t* __restrict__ fixed_pos,
                                                       const uint8_t* __restrict__ mask, int32_t* __restrict__ out_pos,
                                                       uint8_t* __restrict__ out_ok) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= cfg.B) return;
    if (mask && !mask[b]) return;
    const int n = cfg.n_agents, H = cfg.H;
    uint8_t* g = st.grid + (size_t)b * cfg.cells_stride;
    uint64_t* recs = st.agents + (size_t)b * n;
    const bool is_agent = what < 0;
    const int k = -(what + 1);
    if (is_agent) {   // off the grid while a cell is looked for
        uint64_t r = recs[k];
        recs[k] = rec_set(r, MG_AG_FLAGS, rec_byte(r, MG_AG_FLAGS) & ~(MG_AF_PLACED | MG_AF_ACTIVE));
    }
    Mt mt{st.mt + (size_t)b * MG_MT_N, st.mt_pos[b]};
    bool ok = false;
    int x = -1, y = -1;
    const int tries = fixed_pos ? 1 : max_tries;
    for (int t = 0; t < tries && !ok; t++) {
        if (fixed_pos) { x = fixed_pos[2 * b]; y = fixed_pos[2 * b + 1]; if (x < 0 || x >= cfg.W || y < 0 || y >= H) break; }
        else {
            x = x0 + (int)mt.bounded((uint32_t)(x1 - x0 - 1));
            y = y0 + (int)mt.bounded((uint32_t)(y1 - y0 - 1));
        }
        const uint32_t base = g[x * H + y];
        const uint32_t xy = (uint32_t)x | ((uint32_t)y << 8);
        int cnt = 0;
        for (int j = 0; j < n; j++) {
            const uint64_t rj = recs[j];
            cnt += ((rec_byte(rj, MG_AG_FLAGS) & MG_AF_PLACED) && rec_xy(rj) == xy) ? 1 : 0;
        }
        if (!is_agent) {
            // only an empty cell (no object, no agent) accepts a non-agent object (base.py:672-679)
            if (base == 0 && cnt == 0) { g[x * H + y] = (uint8_t)what; ok = true; }
        } else if ((base == 0 || (cfg.obj[base].flags & MG_OF_CAN_OVERLAP)) && (cnt == 0 || (cfg.ghost_mode & 2))) {
            uint64_t r = recs[k];
            const uint32_t old_rank = rec_byte(r, MG_AG_RANK);
            for (int j = 0; j < n; j++) {
                const uint64_t rj = recs[j];
                const uint32_t rk = rec_byte(rj, MG_AG_RANK);
                if (rk > old_rank) recs[j] = rec_set(rj, MG_AG_RANK, rk - 1);
            }
            r = rec_set(r, MG_AG_RANK, (uint32_t)(n - 1));
            r = rec_set(r, MG_AG_X, (uint32_t)x);
            r = rec_set(r, MG_AG_Y, (uint32_t)y);
            r = rec_set(r, MG_AG_FLAGS, (rec_byte(r, MG_AG_FLAGS) & MG_AF_DONE) | MG_AF_ACTIVE | MG_AF_PLACED);
            recs[k] = r;
            ok = true;
        }
    }
    st.mt_pos[b] = mt.pos;
    if (out_pos) { out_pos[2 * b] = ok ? x : -1; out_pos[2 * b + 1] = ok ? y : -1; }
    if (out_ok) out_ok[b] = ok ? 1 : 0;
    if (!ok && !fixed_pos && st.error[b] == 0) st.error[b] = MG_ERR_RECURSION;
}

hipError_t launch_place(const MgConfig& cfg, const MgState& st, int what, int x0, int y0, int x1, int y1, int max_tries,
                        const int32_t* fixed_pos, const uint8_t* mask, int32_t* out_pos, uint8_t* out_ok,
                        hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(place_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cfg, st, what, x0, y0, x1,
                       y1, max_tries, fixed_pos, mask, out_pos, out_ok);
    return hipGetLastError();
}

hipError_t launch_reset(const MgConfig& cfg, const MgState& st, const MgGenProgram& prog, const uint8_t* mask,
                        hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    size_t lds = (size_t)cfg.n_agents * kBlock * sizeof(uint64_t);
    hipLaunchKernelGGL(reset_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), lds, s, cfg, st, prog, mask);
    return hipGetLastError();
}

hipError_t launch_put_obj(const MgConfig& cfg, const MgState& st, int obj, int x, int y, const uint8_t* mask,
                          hipStream_t s) {
    if (cfg.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(put_obj_kernel, dim3((cfg.B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cfg, st, obj, x, y,
                       mask);
    return hipGetLastError();
}

}  // namespace mg
